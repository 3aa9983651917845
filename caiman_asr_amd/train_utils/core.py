"""One training micro-step: autocast forward + loss, cross-rank NaN agreement, backward.
Mirror of training/caiman_asr_train/train_utils/core.py:20-88 and
training/caiman_asr_train/rnnt/model_forward.py:19-101 (bf16 autocast instead of fp16 +
GradScaler: bf16 needs no loss scaling)."""
import contextlib
from argparse import Namespace
from typing import Optional, Tuple

import torch

from caiman_asr_amd.rnnt.loss import IDENTITY_LOSS_MODIFIERS, LossModifiers, get_packing_meta_data


def unwrap(model):
    return getattr(model, "module", model)


def model_loss_forward(model, loss_fn, feats, feat_lens, txt, txt_lens, rnnt_state, loss_mods):
    meta = get_packing_meta_data(feat_lens=feat_lens, txt_lens=txt_lens,
                                 enc_time_reduction=unwrap(model).enc_stack_time_factor, device=feats.device)
    feat_lens = feat_lens.to(feats.device, non_blocking=True)
    txt_lens = txt_lens.to(feats.device, non_blocking=True)
    logits, logits_lens, new_state = model(
        feats, feat_lens, txt, txt_lens, batch_offset=meta["batch_offset"], packed_batch=meta["packed_batch"],
        enc_state=rnnt_state.enc_state if rnnt_state else None,
        pred_net_state=rnnt_state.pred_net_state if rnnt_state else None)
    loss = loss_fn(logits, logits_lens, txt, txt_lens, meta["batch_offset"], meta["max_f_len"], loss_mods=loss_mods)
    return loss, new_state


def model_loss_forward_train(model, loss_fn, args, feats, feat_lens, txt, txt_lens, rnnt_state, loss_mods):
    loss, new_state = model_loss_forward(model, loss_fn, feats, feat_lens, txt, txt_lens, rnnt_state, loss_mods)
    return loss / args.grad_accumulation_batches, new_state


@torch.no_grad()
def model_loss_forward_val(model, loss_fn, feats, feat_lens, txt, txt_lens):
    return model_loss_forward(model, loss_fn, feats, feat_lens, txt, txt_lens, None, IDENTITY_LOSS_MODIFIERS)[0]


def is_loss_nan(loss: torch.Tensor, num_gpus: int) -> bool:
    """All ranks must agree to skip a batch, otherwise the gradient exchange dead-locks
    (core.py:20-42).  One 4-byte MAX all-reduce of an isnan flag instead of an all-gather."""
    flag = torch.isnan(loss).any().to(torch.float32)
    if num_gpus > 1 and torch.distributed.is_initialized():
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
    return bool(flag.item())


def sync_context(model, final_backward: bool):
    """Data-parallel runs exchange gradients ONCE per optimiser step (train_utils/distributed.py): every backward pass
    but the last accumulates only.  With no reducer attached (single GPU) this is a no-op."""
    reducer = getattr(unwrap(model), "grad_reducer", None)
    if reducer is None or final_backward:
        return contextlib.nullcontext()
    return reducer.no_sync()


def train_step(model, loss_fn, args: Namespace, feats, feat_lens, txt, txt_lens, scaler, rnnt_state,
               loss_mods: LossModifiers, final_backward: bool = True) -> Tuple[float, bool, Optional[object]]:
    """-> (loss value, loss_nan, new RNNTState|None).  `scaler`: a torch GradScaler under fp16 autocast
    (`args.amp_dtype = torch.float16`), None under bf16 (the default) or `--no_amp`.  `final_backward` (an extra over
    the reference signature): False for every micro-batch of a gradient-accumulation window except the last."""
    amp = not getattr(args, "no_amp", False)
    amp_dtype = getattr(args, "amp_dtype", torch.bfloat16)
    with torch.autocast("cuda", dtype=amp_dtype, enabled=amp):
        loss, new_state = model_loss_forward_train(model, loss_fn, args, feats, feat_lens, txt, txt_lens,
                                                   rnnt_state, loss_mods)
    loss_nan = is_loss_nan(loss, getattr(args, "num_gpus", 1))
    if not loss_nan:
        with sync_context(model, final_backward):
            if scaler is not None:
                scaler.scale(loss).backward()
            else:
                loss.backward()
    return loss.item(), loss_nan, new_state
