"""N-gram LM hook for beam search (training/caiman_asr_train/lm/kenlm_ngram.py:10-48).

kenlm is a third-party C++ library the reference links through its Python binding; it is not part of this hot
path and is not vendored.  The wrapper binds it when it is importable and otherwise fails loudly -- the beam
search only relies on `begin_state()` and `score_ngram(piece, state) -> (natural-log score, next state)`.
"""
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np


@dataclass
class NgramInfo:
    path: str
    scale_factor: float


class KenLmModel:
    lm_score_scale = 1.0 / np.log10(np.e)  # kenlm reports log10

    def __init__(self, model_path: str):
        try:
            import kenlm
        except ImportError as e:  # pragma: no cover - depends on the deployment image
            raise ImportError("n-gram rescoring needs the `kenlm` Python binding, which is not installed") from e
        self._kenlm = kenlm
        self.model = kenlm.Model(model_path)

    def begin_state(self):
        st = self._kenlm.State()
        self.model.BeginSentenceWrite(st)
        return st

    def score_ngram(self, ngram: str, current_lm_state) -> Tuple[float, object]:
        nxt = self._kenlm.State()
        return self.model.BaseScore(current_lm_state, ngram, nxt) * self.lm_score_scale, nxt


def find_ngram_path(base_path: str) -> Optional[str]:
    """`ngram.binary` is preferred over `ngram.arpa` (kenlm_ngram.py:39-48)."""
    for name in ("ngram.binary", "ngram.arpa"):
        p = os.path.join(base_path, name)
        if os.path.exists(p):
            return p
    return None
