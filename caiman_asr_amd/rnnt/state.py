"""Persistent LSTM state carried between batches (random state passing) and between frames
(streaming decode).  Same fields as training/caiman_asr_train/rnnt/state.py:13-38; selection
helpers follow training/caiman_asr_train/train_utils/rsp.py:108-205."""
from dataclasses import dataclass
from typing import Optional, Tuple

import torch


@dataclass
class EncoderState:
    pre_rnn: Tuple[torch.Tensor, torch.Tensor]   # (h, c) each [pre_rnn_layers, B, H]
    post_rnn: Tuple[torch.Tensor, torch.Tensor]  # (h, c) each [post_rnn_layers, B, H]


@dataclass
class PredNetState:
    next_to_last_pred_state: Tuple[torch.Tensor, torch.Tensor]  # (h, c) each [layers, B, H]
    last_token: torch.Tensor                                    # [B, 1] int


@dataclass
class RNNTState:
    enc_state: EncoderState
    pred_net_state: PredNetState


def _rows_selectable(t):
    """[L, T, B, H] with the rows of a step contiguous and 4-byte granular: what the gather kernel takes"""
    return (t.dim() == 4 and t.stride(3) == 1 and t.stride(2) == t.shape[3]
            and (t.shape[3] * t.element_size()) % 4 == 0 and (t.stride(0) * t.element_size()) % 4 == 0
            and (t.stride(1) * t.element_size()) % 4 == 0 and t.data_ptr() % 4 == 0 and t.shape[0] <= 65535
            and t.shape[1] >= 1 and t.shape[2] >= 1)


def get_last_nonpadded_states(all_hid, lens, how_far_back: int = 0):
    """all_hid: (h, c) each [L, T, B, H]; pick step lens[b]-1-how_far_back per utterance."""
    h, c = all_hid
    if h.is_cuda and _rows_selectable(h) and _rows_selectable(c) and h.shape == c.shape and h.dtype == c.dtype:
        # one launch for both tensors and every layer (include/caiman_rnnt.h caiman_lstm_last_states) instead of the
        # index arithmetic + two gathers below (9 small kernels per stack).  The result is outside autograd: a carried
        # state is detached where it is consumed (training/lib/src/rnnt_ext/custom_lstm/lstm.py:376-377 there,
        # CustomLSTM.forward here), so no gradient ever flows through this selection.
        from caiman_asr_amd import _lib

        h, c = h.detach(), c.detach()
        L, T, B, H = h.shape
        lens_d = lens.to(h.device)
        if lens_d.dtype not in (torch.int32, torch.int64):
            lens_d = lens_d.long()
        lens_d = lens_d.contiguous()
        es = h.element_size()
        h_out, c_out = h.new_empty((L, B, H)), c.new_empty((L, B, H))
        _lib.check(_lib.lib().caiman_lstm_last_states(
            _lib.ptr(h), _lib.ptr(c), L, T, B, H * es, h.stride(0) * es, h.stride(1) * es, c.stride(0) * es, c.stride(1) * es,
            _lib.ptr(lens_d), int(lens_d.dtype == torch.int64), how_far_back, _lib.ptr(h_out), _lib.ptr(c_out), _lib.stream()))
        return h_out, c_out
    idx = (lens.long() - 1 - how_far_back)
    cols = torch.arange(len(lens), device=idx.device)
    return h[:, idx, cols, :], c[:, idx, cols, :]


def maybe_get_last_nonpadded(all_hid, lens):
    return None if all_hid is None else get_last_nonpadded_states(all_hid, lens)


def get_pred_net_state(y, all_pred_hid, y_lens, g_lens) -> Optional[PredNetState]:
    """Last token + the prediction-net state ONE step before the end (rsp.py:132-205): feeding
    (last_token, that state) reproduces the state sequence of the concatenated utterances."""
    if all_pred_hid is None:
        return None
    rows = torch.arange(len(y_lens), device=y.device)
    last_tokens = y[rows, y_lens.long() - 1].unsqueeze(1)
    return PredNetState(
        next_to_last_pred_state=get_last_nonpadded_states(all_pred_hid, g_lens, how_far_back=1),
        last_token=last_tokens)
