"""Transcript -> sentencepiece ids (training/caiman_asr_train/data/tokenizer.py:24-92): word by word, optional
sampling of alternative segmentations with probability `sampling`, id 0 (<unk>) retried and then reported."""
from typing import List, Union

import numpy as np


class Tokenizer:
    def __init__(self, labels: List[str], sentpiece_model: str, sampling: float = 0.0, unk_handling: str = "fail"):
        import sentencepiece as spm

        self.charset = labels
        self.sentpiece_model = sentpiece_model
        self.sampling = sampling
        self.unk_handling = unk_handling      # "fail" | "warn" | "ignore"
        self.sentpiece = spm.SentencePieceProcessor(model_file=sentpiece_model)
        self.num_labels = len(self.sentpiece)

    def _tokenize_word(self, word: str) -> List[int]:
        sample = self.sampling > 0.0 and np.random.random_sample() < self.sampling
        for _ in range(5):   # sampling can split a user-defined symbol into <unk>: try again
            ids = self.sentpiece.encode(word, out_type=int, enable_sampling=bool(sample))
            if 0 not in ids:
                return ids
        if self.unk_handling == "fail":
            raise ValueError(f"<unk> token found in the tokenisation of {word!r}")
        if self.unk_handling == "warn":
            print(f"WARNING: <unk> token found in the tokenisation of {word!r}")
        return ids

    def tokenize(self, transcript: str) -> List[int]:
        return [t for word in transcript.split() for t in self._tokenize_word(word)]

    def detokenize(self, inds: Union[int, List[int]]) -> str:
        if inds == 0:
            return "⁇"    # decode([0]) and decode(0) disagree in sentencepiece; keep one answer
        return self.sentpiece.decode(inds)
