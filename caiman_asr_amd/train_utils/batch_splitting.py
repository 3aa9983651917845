"""Batch-split training step: encoder / prediction at full batch, joint + loss in B/k slices.
Mirror of training/caiman_asr_train/train_utils/batch_splitting.py:19-144.  On 288 GB the logits of a whole
B = 32..128 LibriSpeech batch fit, so `batch_split_factor` is a memory knob here, not a necessity; the step is
kept for configuration parity (`--batch_split_factor`) and is checked to equal the plain step."""
from argparse import Namespace
from typing import Optional, Tuple

import torch

from caiman_asr_amd.rnnt.loss import LossModifiers, get_packing_meta_data
from caiman_asr_amd.train_utils.core import is_loss_nan, sync_context, unwrap


def joint_and_loss(model, loss_fn, args, f, f_lens, g, g_lens, txt, txt_lens, meta_data, loss_mods):
    h = model.joint(f, g, f_lens, g_lens, meta_data["batch_offset"], packed_batch=meta_data["packed_batch"])
    loss = loss_fn(h, f_lens, txt, txt_lens, meta_data["batch_offset"], meta_data["max_f_len"], loss_mods=loss_mods)
    return loss / (args.grad_accumulation_batches * args.batch_split_factor)


def train_step_batch_split(model, loss_fn, args: Namespace, feats, feat_lens, txt, txt_lens, scaler, rnnt_state,
                           loss_mods: LossModifiers, final_backward: bool = True) -> Tuple[float, bool, Optional[object]]:
    """The joint's parameters receive `batch_split_factor` gradient contributions per call, the encoder's and the
    prediction network's one: under data parallelism the slice loop therefore only accumulates, and the joint's
    gradients are declared final once it has run (the reference needs three DDP wrappers with static graphs for the
    same reason, rnnt/sub_models.py:66-79)."""
    m = unwrap(model)
    k = args.batch_split_factor
    batch_size = len(feat_lens)
    assert batch_size % k == 0, "batch size must be divisible by batch_split_factor"
    bs = batch_size // k
    amp = not getattr(args, "no_amp", False)
    amp_dtype = getattr(args, "amp_dtype", torch.bfloat16)
    dev = feats.device
    metas = [get_packing_meta_data(feat_lens[i * bs:(i + 1) * bs], txt_lens[i * bs:(i + 1) * bs],
                                   m.enc_stack_time_factor, device=dev) for i in range(k)]
    feat_lens_d, txt_lens_d = feat_lens.to(dev), txt_lens.to(dev)
    with torch.autocast("cuda", dtype=amp_dtype, enabled=amp):
        (f, f_lens), (g, g_lens), new_state = m.enc_pred(
            feats, feat_lens_d, txt, txt_lens_d,
            enc_state=rnnt_state.enc_state if rnnt_state else None,
            pred_net_state=rnnt_state.pred_net_state if rnnt_state else None)
    # cut the graph: the joint is back-propagated per slice into f_2 / g_2, the encoder / prediction once
    f_2, g_2 = f.detach().requires_grad_(True), g.detach().requires_grad_(True)
    loss_item, batch_has_nan = 0.0, False
    for i in range(k):
        sl = slice(i * bs, (i + 1) * bs)
        with torch.autocast("cuda", dtype=amp_dtype, enabled=amp):
            loss = joint_and_loss(m, loss_fn, args, f_2[sl], f_lens[sl], g_2[sl], g_lens[sl], txt[sl], txt_lens_d[sl],
                                  metas[i], loss_mods)
        if is_loss_nan(loss, getattr(args, "num_gpus", 1)):
            batch_has_nan = True
        with sync_context(model, False):
            if scaler is not None:
                scaler.scale(loss).backward()
            else:
                loss.backward()
        loss_item += loss.item()
    reducer = getattr(m, "grad_reducer", None)
    # A NaN slice drops the whole global batch (train.py:279-284; `batch_has_nan` is agreed across ranks by
    # is_loss_nan's all-reduce): its gradients must not be handed to the reducer, whose finish() will not be called for
    # this window -- the encoder / prediction backward below then only accumulates into an arena that the next window
    # zeroes.
    send = final_backward and not batch_has_nan
    with sync_context(model, send):
        if reducer is not None and send:
            reducer.mark_ready(m.joint_net.parameters())
        f.backward(f_2.grad)
        g.backward(g_2.grad)
    return loss_item, batch_has_nan, None if batch_has_nan else new_state
