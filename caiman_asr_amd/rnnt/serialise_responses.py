"""Turns the live beam into per-frame responses.

Behaviour of training/caiman_asr_train/rnnt/serialise_responses.py:11-205: a *final* is emitted as soon as every
hypothesis in the beam shares a prefix of piece strings (the prefix can never change again), the shared prefix is
then cut off every hypothesis; *partials* list the whole beam, best first; the closing response ships whatever
the best hypothesis still holds.
"""
from typing import Callable, Dict, List, Optional, Tuple

from caiman_asr_amd.rnnt.hypothesis import Hypothesis
from caiman_asr_amd.rnnt.response import DecodingResponse, FrameResponses, HypothesisResponse


def _shared_prefix_len(ordered: List[Hypothesis]) -> int:
    """Number of leading s_seq entries (sentinel included) common to all hypotheses of a list ordered by
    s_seq: the lexicographic extremes differ at the first position where any two hypotheses differ."""
    lo, hi = ordered[0].s_seq, ordered[-1].s_seq
    n, k = min(len(lo), len(hi)), 1
    while k < n and lo[k] == hi[k]:
        k += 1
    return k


class ResponseSerializer:
    def __init__(self, nbest_sort: Callable) -> None:
        self.nbest_sort = nbest_sort

    def frame_responses(self, kept_hyps: Dict[int, Hypothesis], time_idx: Optional[int] = None,
                        partials=True) -> Tuple[FrameResponses, Dict[int, Hypothesis]]:
        final, kept_hyps = self._get_final(kept_hyps)
        part = None
        if partials:
            assert time_idx is not None, "time_idx must be provided if partials is True"
            part = self._build_partials(kept_hyps, time_idx)
        return FrameResponses(partials=part, final=final), kept_hyps

    def last_frame_response(self, kept_hyps: Dict[int, Hypothesis]) -> FrameResponses:
        best = self.nbest_sort(kept_hyps.values())[0]
        final = self._build_final([best], len(best.y_seq)) if len(best.y_seq) > 1 else None
        return FrameResponses(partials=None, final=final)

    def _build_partials(self, kept_hyps: Dict[int, Hypothesis], time_idx: int) -> DecodingResponse:
        alts, start = [], time_idx
        for hyp in self.nbest_sort(kept_hyps.values()):
            if len(hyp.timesteps) <= 1:
                continue  # nothing beyond the sentinel
            start = min(start, min(hyp.timesteps[1:]))
            alts.append(HypothesisResponse(y_seq=hyp.y_seq[1:], timesteps=hyp.timesteps[1:], token_seq=hyp.s_seq[1:],
                                           confidence=hyp.p_seq[1:]))
        return DecodingResponse(start_frame_idx=start, duration_frames=time_idx - start + 1, is_provisional=True,
                                alternatives=alts)

    def _get_final(self, kept_hyps: Dict[int, Hypothesis]):
        # ordered by piece strings: the first one also supplies the confidences of the final (:128,:176)
        hyps = sorted(kept_hyps.values(), key=lambda h: h.s_seq)
        k = _shared_prefix_len(hyps)
        if k == 1:
            return None, kept_hyps
        final = self._build_final(hyps, k)
        for h in hyps:
            h.truncate(k)
        return final, kept_hyps

    def _build_final(self, hyps: List[Hypothesis], tkn_idx: int) -> DecodingResponse:
        head = hyps[0]
        for h in hyps[1:]:
            assert h.s_seq[1:tkn_idx] == head.s_seq[1:tkn_idx], "finals must share their piece strings"
            assert h.y_seq[1:tkn_idx] == head.y_seq[1:tkn_idx], "finals must share their token ids"
            assert len(h.timesteps[1:tkn_idx]) == len(head.timesteps[1:tkn_idx])
        # a token spoken with confidence has been spoken by the earliest frame any hypothesis saw it
        frames = [min(h.timesteps[i] for h in hyps) for i in range(1, min(tkn_idx, len(head.timesteps)))]
        resp = HypothesisResponse(y_seq=head.y_seq[1:tkn_idx], timesteps=frames, token_seq=head.s_seq[1:tkn_idx],
                                  confidence=head.p_seq[1:tkn_idx])
        return DecodingResponse(start_frame_idx=min(frames), duration_frames=max(frames) - min(frames) + 1,
                                is_provisional=False, alternatives=[resp])
