"""Validation pass: batches -> decoder -> transcripts -> error rate
(training/caiman_asr_train/evaluate/core.py:134-413, reduced to the part that sits on the hot path: encode + search
on the device, detokenise and score on the host).  Works with any decoder of `caiman_asr_amd.rnnt` (greedy, beam,
native beam): they share `decode(feats, feat_lens) -> [{frame: FrameResponses}]`."""
from typing import Callable, Dict, Iterable, List, Optional

import torch

from caiman_asr_amd.evaluate import state_resets
from caiman_asr_amd.evaluate.metrics import ErrorRate, word_error_rate
from caiman_asr_amd.rnnt.decoder import flatten_responses


@torch.no_grad()
def evaluate(loader: Iterable, decoder, detokenize: Callable[[List[int]], str], error_rate: ErrorRate = ErrorRate.WORD,
             standardize: bool = True, autocast_dtype: Optional[torch.dtype] = torch.bfloat16,
             max_batches: Optional[int] = None, sr_segment: Optional[float] = None, sr_overlap: float = 0.0,
             model_config: Optional[dict] = None, enc_time_reduction: int = 2, eos_idx: Optional[int] = None) -> Dict:
    """loader yields (feats [T,B,F], feat_lens, txt [B,U], txt_lens) -> {"wer", "errors", "words", "hypotheses",
    "references", "timestamps"}.  With `sr_segment` (seconds; `model_config` = the parsed YAML) long utterances are
    decoded in overlapping windows with the state reset at every window start and stitched back together
    (evaluate/state_resets.py; reference: evaluate/core.py:215-240 with --sr_segment / --sr_overlap); `eos_idx`: stop an
    utterance at its first end-of-sequence token (--eos_is_terminal)."""
    hyps: List[str] = []
    refs: List[str] = []
    stamps: List[List[int]] = []
    for i, (feats, f_lens, txt, t_lens) in enumerate(loader):
        if max_batches is not None and i >= max_batches:
            break
        plans = None
        if sr_segment:
            feats, f_lens, plans = state_resets.split_batch(feats, f_lens, sr_segment, sr_overlap, model_config)
        if autocast_dtype is None:
            out = decoder.decode(feats, f_lens)
        else:
            with torch.autocast("cuda", dtype=autocast_dtype):
                out = decoder.decode(feats, f_lens)
        tokens, frames, confs = flatten_responses(out)
        if plans is not None:
            tokens, frames, _ = state_resets.merge_batch(tokens, frames, confs, enc_time_reduction, plans, eos_idx)
        txt_h, len_h = txt.cpu(), t_lens.cpu().tolist()
        for b, tk in enumerate(tokens):
            hyps.append(detokenize(tk))
            refs.append(detokenize(txt_h[b, : len_h[b]].tolist()))
            stamps.append(frames[b])
    wer, errors, words = word_error_rate(hyps, refs, error_rate, standardize)
    return dict(wer=wer, errors=errors, words=words, hypotheses=hyps, references=refs, timestamps=stamps)
