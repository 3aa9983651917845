"""EOS handling strategies (training/caiman_asr_train/rnnt/eos_strategy.py:7-29)."""
from dataclasses import dataclass
from typing import Union


@dataclass
class EOSIgnore:
    eos_idx: int


@dataclass
class EOSBlank:
    eos_idx: int


@dataclass
class EOSPredict:
    eos_idx: int
    alpha: float
    beta: float


EOSStrategy = Union[None, EOSIgnore, EOSPredict, EOSBlank]
