"""Loss wrapper with the reference's calling convention.
Interface mirror of training/caiman_asr_train/rnnt/loss.py:26-173."""
from dataclasses import dataclass
from typing import Optional

import torch

from caiman_asr_amd.rnnt_ext.transducer.loss import TransducerLoss


@dataclass
class LossModifiers:
    delay_penalty: float
    eos_penalty: float
    star_penalty: float


IDENTITY_LOSS_MODIFIERS = LossModifiers(delay_penalty=0.0, eos_penalty=0.0, star_penalty=1.0)


class ApexTransducerLoss(torch.nn.Module):
    """Batch-mean RNN-T loss on (optionally packed) joint logits."""

    def __init__(self, blank_idx: int, eos_idx: Optional[int], star_idx: Optional[int], packed_input: bool,
                 validate_first_n_remaining: int = 10):
        super().__init__()
        self.t_loss = TransducerLoss(packed_input=packed_input)
        self.packed_input = packed_input
        self.blank_idx = blank_idx
        self.eos_idx = eos_idx
        self.star_idx = star_idx
        self.validate_first_n_remaining = validate_first_n_remaining

    def forward(self, logits: torch.Tensor, logit_lens: torch.Tensor, y: torch.Tensor, y_lens: torch.Tensor,
                batch_offset: Optional[torch.Tensor], max_f_len: Optional[int],
                loss_mods: LossModifiers = IDENTITY_LOSS_MODIFIERS):
        if y.dtype != torch.int32:
            y = y.int()
        if logit_lens.dtype != torch.int32:
            logit_lens = logit_lens.int()
        if y_lens.dtype != torch.int32:
            y_lens = y_lens.int()
        if self.validate_first_n_remaining > 0:
            # device->host sync: only the first few calls (loss.py:105-110)
            self._validate_inputs(logits, batch_offset, y)
            self.validate_first_n_remaining -= 1
        return self.t_loss(
            logits, y.contiguous(), logit_lens.contiguous(), y_lens.contiguous(), self.blank_idx,
            eos_idx=self.eos_idx, star_idx=self.star_idx, batch_offset=batch_offset, max_f_len=max_f_len,
            delay_penalty=loss_mods.delay_penalty, eos_penalty=loss_mods.eos_penalty,
            star_penalty=loss_mods.star_penalty).mean()

    def _validate_inputs(self, logits, batch_offset, y) -> None:
        if self.packed_input:
            assert len(logits.shape) == 2, \
                f"When packed_input=True, logits should be of shape [total_packed, K+1] but {logits.shape=}"
            total_packed = logits.shape[0]
            assert total_packed == batch_offset[-1].item(), \
                f"Packed input shape and batch_offsets are inconsistent: {total_packed} != {batch_offset[-1].item()}"
        else:
            assert len(logits.shape) == 4, \
                f"When packed_input=False, logits should be of shape [B, T, U, K+1] but {logits.shape=}"
            assert y.shape[1] == logits.shape[2] - 1, \
                f"When packed_input=False, {y.shape[1]=} should be 1 less than {logits.shape[2]=}"


def get_packing_meta_data(feat_lens: torch.Tensor, txt_lens: torch.Tensor, enc_time_reduction: int,
                          device=None) -> dict:
    """batch_offset = cumsum(ceil(feat_len / reduction) * (txt_len + 1)); max_f_len; packed_batch.

    Same arithmetic as the reference (loss.py:155-173).  When the lengths are HOST tensors the two
    scalars are read without a device sync and `batch_offset` is shipped to `device` asynchronously
    (the reference syncs twice per step here and once more in RNNT.joint)."""
    final_feat_lens = (feat_lens + enc_time_reduction - 1) // enc_time_reduction
    batch_offset = torch.cumsum(final_feat_lens * (txt_lens + 1), dim=0)
    meta = {"max_f_len": int(final_feat_lens.max().item()), "packed_batch": int(batch_offset[-1].item())}
    if device is not None and batch_offset.device != torch.device(device):
        batch_offset = batch_offset.to(device, non_blocking=True)
    meta["batch_offset"] = batch_offset
    return meta
