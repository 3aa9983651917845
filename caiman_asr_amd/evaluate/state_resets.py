"""State resets for long-form evaluation: an utterance longer than `sr_segment` seconds is cut
into overlapping windows that are decoded as independent batch rows -- every window starts from the zero state, which is
the "reset" -- and the per-window hypotheses are stitched back into one.

Mirror of training/caiman_asr_train/evaluate/state_resets/ (core.py:17-403, batch.py:15-163,
overlap_processing.py:15-249, timestamp.py:8-64) at the call sites of evaluate/core.py:215-240; same results on the same
inputs (tests/golden/state_resets.json is produced by the reference's functions), written around one plan object and
`Tensor.unfold` instead of pad / cat / reshape chains.

    feats, feat_lens, plans = split_batch(feats, feat_lens, sr_segment, sr_overlap, cfg)
    ... decode the windows as a batch ...
    tokens, stamps, probs = merge_batch(tokens, stamps, probs, enc_time_reduction, plans, eos_idx)
"""
import math
import warnings
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple, Union

import torch


@dataclass
class FullStamp:
    """model: the frame on which the model emitted the token; user_perceived: the frame after partial -> final overwrites
    (timestamp.py:8-27)."""
    model: int
    user_perceived: int


Timestamp = Union[FullStamp, int]


def model_time(t: Timestamp) -> int:
    return t.model if isinstance(t, FullStamp) else t


def user_perceived_time(t: Timestamp) -> int:
    return t.user_perceived if isinstance(t, FullStamp) else t


def shift(t: Timestamp, n: int) -> Timestamp:
    return FullStamp(t.model + n, t.user_perceived + n) if isinstance(t, FullStamp) else t + n


@dataclass(frozen=True)
class WindowPlan:
    """How one utterance was cut: `n_windows` windows of `window` frames, consecutive windows sharing `overlap` frames."""
    n_windows: int
    window: int
    overlap: int

    @property
    def hop(self) -> int:
        return self.window - self.overlap


def frame_seconds(cfg: dict) -> float:
    """Duration of one (stacked) input frame (utils/frame_width.py:31-58)."""
    stride = cfg["input_train"]["filterbank_features"]["window_stride"]
    splice = cfg["input_train"]["frame_splicing"]
    if splice["frame_stacking"] != splice["frame_subsampling"]:
        raise AssertionError("ERROR: please use the same frame stacking and frame subsampling.")
    return stride * splice["frame_stacking"]


def window_frames(sr_segment: float, sr_overlap: float, cfg: dict) -> Tuple[int, int]:
    """Seconds -> frames, with the reference's argument checks (core.py:123-187)."""
    if sr_segment <= 0 or sr_overlap < 0:
        raise ValueError("Please ensure you provide positive --sr_segment and non-negative --sr_overlap to use State Resets.")
    if sr_segment <= sr_overlap:
        raise ValueError("Please ensure that --sr_segment is greater than --sr_overlap when using State Resets.")
    w = frame_seconds(cfg)
    return round(sr_segment / w), round(sr_overlap / w)


def window_count(n_frames: int, window: int, overlap: int) -> Tuple[int, int]:
    """-> (number of windows, zero frames to append so that the last window is full) (core.py:365-403)."""
    hop = window - overlap
    n, rest = divmod(n_frames - overlap, hop)
    return (n, 0) if rest == 0 else (n + 1, hop - rest)


def split_utterance(feats: torch.Tensor, feat_lens: torch.Tensor, window: int, overlap: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """feats [T, 1, F] -> [window, n_windows, F] (window i = frames [i * hop, i * hop + window) of the zero-padded
    utterance), lens [n_windows] = window.  Shorter than one window: unchanged (core.py:190-256)."""
    if feats.shape[1] != 1:
        raise AssertionError(f"feats with size {feats.size()} are batched, set --val_batch_size=1")
    if feats.shape[0] < window:
        return feats, feat_lens
    n, pad = window_count(int(feat_lens.item()), window, overlap)
    need = (n - 1) * (window - overlap) + window
    x = feats[:, 0]
    if need > x.shape[0]:
        x = torch.nn.functional.pad(x, (0, 0, 0, need - x.shape[0]))
    # unfold: [n, F, window] views of the padded utterance, step = hop
    win = x[:need].unfold(0, window, window - overlap).permute(2, 0, 1).contiguous()
    lens = torch.full((n,), window, dtype=torch.int32, device=feats.device)
    return win, lens


def split_batch(feats: torch.Tensor, feat_lens: torch.Tensor, sr_segment: float, sr_overlap: float,
                cfg: dict) -> Tuple[torch.Tensor, torch.Tensor, List[WindowPlan]]:
    """feats [T, B, F] -> windows of every utterance side by side in the batch dimension, zero-padded to the longest
    (batch.py:15-84).  plans[b] tells merge_batch() how many rows belong to utterance b."""
    window, overlap = window_frames(sr_segment, sr_overlap, cfg)
    cols, lens, plans = [], [], []
    for b in range(feats.shape[1]):
        n = feat_lens[b].unsqueeze(0)
        w, wl = split_utterance(feats[: int(n.item()), b].unsqueeze(1), n, window, overlap)
        cols += [w[:, i] for i in range(w.shape[1])]
        lens += [wl[i] for i in range(w.shape[1])]
        plans.append(WindowPlan(w.shape[1], window, overlap))
    return torch.nn.utils.rnn.pad_sequence(cols), torch.stack(lens), plans


def _drop_repeats(tokens: List[int], stamps: List[Timestamp], probs: Optional[List[float]], trusted: List[int],
                  lookahead: int):
    """A token among the first `lookahead` of a window that also closes the previous window was decoded twice: drop it
    here, and only look for later duplicates after its position in the previous window (overlap_processing.py:196-230).
    The scan walks the ORIGINAL first `lookahead` tokens while the lists shrink, as the reference does."""
    for tok in tokens[:lookahead]:
        if tok in trusted:
            i = tokens.index(tok)
            del stamps[i]
            if probs:
                del probs[i]
            del tokens[i]
            trusted = trusted[trusted.index(tok) + 1:]
    return tokens, stamps, probs


def merge_windows(tokens: Sequence[List[int]], stamps: Sequence[List[Timestamp]], probs: Sequence[List[float]],
                  enc_time_reduction: int, plan_window: int, plan_overlap: int, lookahead: int = 3):
    """The windows of ONE utterance -> ([tokens], [timestamps], [probs] | None) as if decoded in one go
    (core.py:64-121): tokens emitted inside the overlap of a window are dropped, then repeats across the boundary,
    and window i's timestamps move forward by i * hop encoder frames."""
    overlap_enc = math.ceil(plan_overlap / enc_time_reduction)
    have_probs = bool(probs)
    kept_t, kept_s, kept_p = [list(tokens[0])], [list(stamps[0])], [list(probs[0])] if have_probs else None
    for i in range(1, len(tokens)):
        skip = 0
        for t in stamps[i]:
            if model_time(t) >= overlap_enc:
                break
            skip += 1
        kept_t.append(list(tokens[i][skip:]))
        kept_s.append(list(stamps[i][skip:]))
        if have_probs:
            kept_p.append(list(probs[i][skip:]))
    for i in range(1, len(kept_t)):
        kept_t[i], kept_s[i], p = _drop_repeats(kept_t[i], kept_s[i], kept_p[i] if kept_p else None,
                                                kept_t[i - 1][-lookahead:], lookahead)
        if have_probs:
            kept_p[i] = p
    hop = plan_window - plan_overlap
    if hop % enc_time_reduction != 0:
        warnings.warn(f"segment_frames={plan_window} - overlap_frames={plan_overlap} must be divisible by "
                      f"enc_time_reduction={enc_time_reduction} in order to have accurate integer timestamps")
    step = hop // enc_time_reduction
    flat_s = [shift(t, i * step) if i else t for i, row in enumerate(kept_s) for t in row]
    flat_t = [[t for row in kept_t for t in row]]
    flat_p = [[p for row in kept_p for p in row]] if have_probs else kept_p
    return flat_t, [flat_s], flat_p


def merge_batch(tokens: List[List[int]], stamps: List[List[Timestamp]], probs: List[List[float]], enc_time_reduction: int,
                plans: Sequence[WindowPlan], eos_idx: Optional[int] = None):
    """Rows of split_batch()'s batch -> one hypothesis per original utterance (batch.py:87-163).  With `eos_idx` the
    windows behind the first one that contains the end-of-sequence token are ignored."""
    out_t, out_s, out_p = [], [], []
    row = 0
    group_probs = None
    for plan in plans:
        t, s, p = tokens[row: row + plan.n_windows], stamps[row: row + plan.n_windows], probs[row: row + plan.n_windows]
        if eos_idx is not None:
            last = next((j for j, w in enumerate(t) if eos_idx in w), len(t) - 1)
            t, s, p = t[: last + 1], s[: last + 1], p[: last + 1]
        mt, ms, group_probs = merge_windows(t, s, p, enc_time_reduction, plan.window, plan.overlap)
        out_t += mt
        out_s += ms
        if group_probs:
            out_p += group_probs
        row += plan.n_windows
    assert len(out_t) == len(plans)
    return out_t, out_s, (out_p if group_probs else None)
