"""Step schedules for the loss modifiers: delay penalty 0 -> 0.01 and star penalty 0.75 -> 1.0, switched at a fixed
step or when the dev WER drops under a threshold (training/caiman_asr_train/train_utils/schedule.py:7-114; wired
up in setup/train.py:212-229).  The values feed `LossModifiers` -> caiman_transducer_loss_forward / _backward."""
from abc import ABC, abstractmethod
from typing import Any, Dict, Optional


class Schedule(ABC):
    @abstractmethod
    def step(self, train_step: int, *, hints: Optional[Dict[str, Any]] = None) -> float:
        """Advance to `train_step` (`hints`, e.g. {"wer": best_wer}) and return the value."""

    @abstractmethod
    def value(self) -> float:
        """Value at the last step() (initial value before any)."""


class ConstantSchedule(Schedule):
    def __init__(self, value: float) -> None:
        self._value = float(value)

    def step(self, train_step: int, *, hints: Optional[Dict[str, Any]] = None) -> float:
        return self._value

    def value(self) -> float:
        return self._value


class StepSchedule(Schedule):
    """initial_value until `train_step >= toggle_step` or `hints["wer"] < wer_threshold`, final_value from then on
    (latched: it never switches back)."""

    def __init__(self, initial_value: float, final_value: float = 1.0, toggle_step: Optional[int] = None,
                 wer_threshold: Optional[float] = None) -> None:
        if toggle_step is None and wer_threshold is None:
            raise ValueError("StepSchedule is not set to change at any step or WER threshold")
        self.initial_value, self.final_value = float(initial_value), float(final_value)
        self.toggle_step, self.wer_threshold = toggle_step, wer_threshold
        self.set = False

    def step(self, train_step: int, *, hints: Optional[Dict[str, Any]] = None) -> float:
        if not self.set:
            wer = None if hints is None else hints.get("wer")
            if self.wer_threshold is not None and self.toggle_step is None and wer is None:
                raise ValueError("StepSchedule expecting WER in hints but it was not found.")
            by_wer = self.wer_threshold is not None and wer is not None and wer < self.wer_threshold
            by_step = self.toggle_step is not None and train_step >= self.toggle_step
            self.set = by_wer or by_step
        return self.value()

    def value(self) -> float:
        return self.final_value if self.set else self.initial_value


def build_delay_penalty_scheduler(args) -> Schedule:
    """`--delay_penalty` "wer_schedule" (default: 0 -> 0.01 at dev WER < 0.3 or `--dp_toggle_step`) or a constant
    (setup/train.py:220-229; defaults args/delay_penalty.py:12-45)."""
    dp = getattr(args, "delay_penalty", "wer_schedule")
    if dp == "wer_schedule":
        return StepSchedule(getattr(args, "dp_initial_value", 0.0), getattr(args, "dp_final_value", 0.01),
                            toggle_step=getattr(args, "dp_toggle_step", None),
                            wer_threshold=getattr(args, "dp_wer_threshold", 0.3))
    return ConstantSchedule(float(dp))


def build_star_scheduler(args) -> StepSchedule:
    """Star penalty 0.75 -> 1.0 at dev WER < 0.2 or `--star_toggle_step` (setup/train.py:212-218; args/star.py)."""
    return StepSchedule(getattr(args, "star_initial_value", 0.75), getattr(args, "star_final_value", 1.0),
                        toggle_step=getattr(args, "star_toggle_step", None),
                        wer_threshold=getattr(args, "star_wer_threshold", 0.2))
