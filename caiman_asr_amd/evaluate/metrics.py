"""Word / character / mixture error rate: training/caiman_asr_train/evaluate/metrics.py:81-166,
error_rates.py:12-73, data/text/has_spaces.py.

`standardize=True` in the reference runs its training-time cleaner (number spelling via `inflect`) and a vendored
Whisper English normaliser over both sides first; those text-normalisation subsystems are not restated here.  This
module standardises with what is cheap and unambiguous -- lower-casing, dropping characters outside
[a-z ' space], removing <tags> and collapsing blanks -- which coincides with the reference on transcripts that are
already normalised (its own test table); for anything else: parity unpinned, pass pre-normalised text.
"""
import re
import string
import unicodedata
from enum import Enum
from typing import Dict, List, Tuple

import numpy as np

from caiman_asr_amd import _lib


class ErrorRate(Enum):
    WORD = 1
    CHAR = 2
    MIXTURE = 3


def get_error_rate(cfg: dict) -> ErrorRate:
    name = cfg["error_rate"].lower()
    table = {"wer": ErrorRate.WORD, "word": ErrorRate.WORD, "cer": ErrorRate.CHAR, "char": ErrorRate.CHAR,
             "mer": ErrorRate.MIXTURE, "mixture": ErrorRate.MIXTURE}
    if name not in table:
        raise ValueError(f"Invalid error rate: {cfg['error_rate']}")
    return table[name]


def error_rate_abbrev(error_rate: ErrorRate) -> str:
    return {ErrorRate.WORD: "wer", ErrorRate.CHAR: "cer", ErrorRate.MIXTURE: "mer"}[error_rate]


def _is_cjk(ch: str) -> bool:
    return unicodedata.name(ch, "").startswith("CJK ")


def decide_and_split(text: str, error_rate: ErrorRate) -> List[str]:
    """Words; characters; or words with every CJK character a token of its own."""
    if error_rate is ErrorRate.WORD:
        return text.split()
    if error_rate is ErrorRate.CHAR:
        return " ".join(text).split()
    if error_rate is ErrorRate.MIXTURE:
        return "".join(f" {c} " if _is_cjk(c) else c for c in text).split()
    raise ValueError(f"Invalid error rate: {error_rate}")


def levenshtein(a: List, b: List) -> int:
    """Edit distance between two token lists, computed by the library (`caiman_levenshtein`)."""
    ids: Dict = {}
    ia = np.fromiter((ids.setdefault(t, len(ids)) for t in a), dtype=np.int32, count=len(a))
    ib = np.fromiter((ids.setdefault(t, len(ids)) for t in b), dtype=np.int32, count=len(b))
    d = _lib.lib().caiman_levenshtein(ia.ctypes.data if len(a) else None, len(a), ib.ctypes.data if len(b) else None, len(b))
    if d < 0:
        raise RuntimeError(_lib.lib().caiman_last_error().decode())
    return int(d)


_KEEP = set(string.ascii_lowercase + " '")


def standardize_wer(text: str) -> str:
    text = re.sub(r"<[^>]*>", " ", text.lower())
    return " ".join("".join(c if c in _KEEP else " " for c in text).split())


def word_error_rate(hypotheses: List[str], references: List[str], error_rate: ErrorRate = ErrorRate.WORD,
                    standardize: bool = True) -> Tuple[float, int, int]:
    """-> (error rate, edit operations, reference tokens); inf when the references hold no token."""
    if len(references) != len(hypotheses):
        raise ValueError(f"Unequal number of hypotheses and references: {len(hypotheses)} and {len(references)}")
    scores = words = 0
    for hyp, ref in zip(hypotheses, references):
        if standardize:
            hyp, ref = standardize_wer(hyp), standardize_wer(ref)
        h, r = decide_and_split(hyp, error_rate), decide_and_split(ref, error_rate)
        words += len(r)
        scores += levenshtein(h, r)
    return (scores / words if words else float("inf")), scores, words
