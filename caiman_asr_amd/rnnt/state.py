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


class BatchChunks:
    """All hidden states of an LSTM stack, kept as one (h, c) pair [L, T, b, H] per chunk of the batch (rnnt/model.py runs the
    layer pipeline 32 utterances at a time where the weight-resident kernels only exist for up to 32): the selections below
    gather from each chunk and concatenate the small results, instead of concatenating gigabytes of states first."""

    def __init__(self, parts, bounds):
        self.parts, self.bounds = list(parts), list(bounds)


def get_last_nonpadded_states(all_hid, lens, how_far_back: int = 0):
    """all_hid: (h, c) each [L, T, B, H]; pick step lens[b]-1-how_far_back per utterance."""
    if isinstance(all_hid, BatchChunks):
        got = [get_last_nonpadded_states(part, lens[a:b], how_far_back) for part, (a, b) in zip(all_hid.parts, all_hid.bounds)]
        return torch.cat([g[0] for g in got], dim=1), torch.cat([g[1] for g in got], dim=1)
    idx = (lens.long() - 1 - how_far_back)
    cols = torch.arange(len(lens), device=idx.device)
    return all_hid[0][:, idx, cols, :], all_hid[1][:, idx, cols, :]


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
