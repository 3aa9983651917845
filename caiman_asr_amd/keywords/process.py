"""Keyword list loader: JSON `{"keywords": {"<phrase>": <weight>, ...}}`, spaces become the sentencepiece
word-boundary mark (training/caiman_asr_train/keywords/process.py:9-39)."""
import json
from numbers import Number

from caiman_asr_amd.keywords.trie import Keywords

SPACE_MARK = "▁"


def load_keywords(path: str) -> Keywords:
    with open(path, "r") as f:
        doc = json.load(f)
    kw = doc.get("keywords") if isinstance(doc, dict) else None
    if not isinstance(kw, dict) or set(doc) != {"keywords"}:
        raise ValueError(f"Schema not matched: expected a single 'keywords' mapping in {path}")
    for k, v in kw.items():
        if not isinstance(k, str) or isinstance(v, bool) or not isinstance(v, Number):
            raise ValueError(f"Schema not matched: {k!r}: {v!r} is not a string -> number entry")
    return Keywords([(k.replace(" ", SPACE_MARK), float(v)) for k, v in kw.items()])
