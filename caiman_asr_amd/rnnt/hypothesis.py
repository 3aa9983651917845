"""Beam-search hypothesis record, field-compatible with
training/caiman_asr_train/rnnt/hypothesis.py:38-162.

A hypothesis carries the running log-score, the non-blank tokens / frames / piece strings / confidences that
have not yet been shipped as a final (element 0 is always a sentinel: SOS, or the last shipped token), and a
rolling hash of the *text* so that different tokenisations of the same string merge (hypothesis.py:114-122:
h <- (h * 0x10FFFF + ord(c)) mod 1_000_000_000_039 per character).
"""
from typing import Dict, List, Optional, Tuple

import torch

SPU_UNICODE = 0x2581
CHR_SPU_UNICODE = chr(SPU_UNICODE)
MAX_UNICODE = 0x10FFFF
HASHSIZE = 1_000_000_000_039


def roll_hash(h: int, text: str) -> int:
    for ch in text:
        h = (h * MAX_UNICODE + ord(ch)) % HASHSIZE
    return h


def token_strs_to_transcript(tokens: List[str]) -> str:
    return "".join(tokens).replace(CHR_SPU_UNICODE, " ").strip()


class Hypothesis:
    __slots__ = ("score", "p_seq", "y_seq", "y_len_t", "timesteps", "s_seq", "hashval", "pred_state",
                 "ngram_lm_state", "is_terminal", "kws_state", "_prev_length")

    def __init__(self, score: float, p_seq: List[float], y_seq: List[int], y_len_t: int, timesteps: List[int],
                 s_seq: List[str], hashval: int, pred_state: Optional[Tuple[torch.Tensor, torch.Tensor]],
                 ngram_lm_state=None, is_terminal: bool = False, kws_state: Optional[Dict[int, float]] = None,
                 _prev_length: int = 0):
        self.score = score
        self.p_seq = p_seq
        self.y_seq = y_seq
        self.y_len_t = y_len_t
        self.timesteps = timesteps
        self.s_seq = s_seq
        self.hashval = hashval
        self.pred_state = pred_state
        self.ngram_lm_state = ngram_lm_state
        self.is_terminal = is_terminal
        self.kws_state = {0: 0.0} if kws_state is None else kws_state
        self._prev_length = _prev_length

    def __repr__(self):
        return f"Hypothesis(score={self.score:6.2f}, '{self.transcript}')"

    @property
    def y_last(self) -> int:
        return self.y_seq[-1]

    @property
    def y_length_tot(self) -> int:
        """Non-blank tokens so far, including those already shipped (the sentinel counts once)."""
        return len(self.y_seq) + self._prev_length

    @property
    def transcript(self) -> str:
        return token_strs_to_transcript(self.s_seq[1:])

    def truncate(self, tkn_idx: int) -> None:
        """Drop everything before token `tkn_idx - 1`, which stays as the new sentinel."""
        cut = tkn_idx - 1
        self._prev_length += cut
        self.p_seq = self.p_seq[cut:]
        self.s_seq = self.s_seq[cut:]
        self.y_seq = self.y_seq[cut:]
        self.timesteps = self.timesteps[cut:]

    def update_hash(self, new_str: str) -> None:
        self.hashval = roll_hash(self.hashval, new_str)

    def clone(self) -> "Hypothesis":
        """Copy the sequences; share the (immutable) prediction / LM states."""
        return Hypothesis(self.score, list(self.p_seq), list(self.y_seq), self.y_len_t, list(self.timesteps),
                          list(self.s_seq), self.hashval, self.pred_state, self.ngram_lm_state, self.is_terminal,
                          dict(self.kws_state), self._prev_length)

    def check(self) -> None:
        assert len(self.y_seq) > 0 and self.y_length_tot >= len(self.y_seq)
        assert len(self.y_seq) == len(self.timesteps) == len(self.s_seq) == len(self.p_seq)


def init_sos_hyp(sos_tkn: int, ngram_lm=None) -> Hypothesis:
    """The empty hypothesis every search starts from (hypothesis.py:169-189)."""
    lm_state = ngram_lm.begin_state() if ngram_lm is not None else None
    return Hypothesis(score=0.0, p_seq=[1.0], y_seq=[sos_tkn], y_len_t=1, timesteps=[-1], s_seq=[CHR_SPU_UNICODE],
                      hashval=0, pred_state=None, ngram_lm_state=lm_state)
