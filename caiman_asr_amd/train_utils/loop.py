"""The per-batch logic of one optimiser step: gradient accumulation, NaN handling, loss-modifier schedules, random
state passing.  Mirror of the body of the reference's training loop (training/caiman_asr_train/train.py:215-300),
without its logging / validation / checkpoint control plane:

    accumulated == 0          -> adjust LR, zero gradients, step the delay / star schedules
    every micro-batch         -> train_step (or train_step_batch_split); a NaN loss drops the whole global batch
                                 (train.py:279-284); rsp_end_step decides which state is carried on
    accumulated == grad_accumulation_batches -> (data parallel: ONE gradient exchange) optimiser step + EMA

Under data parallelism only the last micro-batch's backward pass is allowed to hand gradients to the reducer
(train_utils/distributed.py): the reference's DDP all-reduces in every backward pass instead.
"""
from argparse import Namespace
from typing import Callable, Optional

from caiman_asr_amd.rnnt.loss import LossModifiers
from caiman_asr_amd.train_utils.batch_splitting import train_step_batch_split
from caiman_asr_amd.train_utils.core import train_step
from caiman_asr_amd.train_utils.rsp import generate_batch_history, rsp_end_step
from caiman_asr_amd.train_utils.schedule import ConstantSchedule, Schedule


class TrainStepper:
    def __init__(self, model, loss_fn, args: Namespace, optimizer_wrapper, adjust_lr: Optional[Callable[[int], None]] = None,
                 dp_scheduler: Optional[Schedule] = None, star_scheduler: Optional[Schedule] = None, scaler=None):
        self.model, self.loss_fn, self.args, self.opt = model, loss_fn, args, optimizer_wrapper
        self.adjust_lr = adjust_lr
        self.dp = dp_scheduler or ConstantSchedule(0.0)
        self.star = star_scheduler or ConstantSchedule(1.0)
        self.scaler = scaler
        self.step_fn = train_step_batch_split if getattr(args, "batch_split_factor", 1) > 1 else train_step
        self.accumulated = 0
        self.losses = []
        self.state = None
        self.step = getattr(args, "start_step", 1)
        seq = getattr(args, "rsp_seq_len_freq", [1])
        self.rsp_counter = generate_batch_history(seq)
        if not hasattr(args, "rsp_seq_len_freq"):
            args.rsp_seq_len_freq, args.rsp_delay = [1], 0

    def micro_batch(self, feats, feat_lens, txt, txt_lens, best_wer: float = float("inf")) -> Optional[dict]:
        """One batch from the loader.  Returns None while the accumulation window is open, else a record of the
        optimiser step that closed it ({"step", "loss", "lr"})."""
        a = self.args
        if self.accumulated == 0:
            if self.adjust_lr is not None:
                self.adjust_lr(self.step)
            self.opt.zero_grad()
            self.losses = []
            self.dp.step(self.step, hints={"wer": best_wer})
            self.star.step(self.step, hints={"wer": best_wer})
        final = self.accumulated + 1 == a.grad_accumulation_batches
        mods = LossModifiers(delay_penalty=self.dp.value(), star_penalty=self.star.value(),
                             eos_penalty=getattr(a, "eos_penalty", 0.0))
        loss_item, loss_nan, state = self.step_fn(self.model, self.loss_fn, a, feats, feat_lens, txt, txt_lens,
                                                  scaler=self.scaler, rnnt_state=self.state, loss_mods=mods,
                                                  final_backward=final)
        if loss_nan:
            self.accumulated = 0   # NaNs pollute the accumulated gradients: the window restarts with zero_grad
            if hasattr(self.opt, "drop_window"):
                self.opt.drop_window()   # data parallel: nothing of the dropped window stays marked / in flight
        else:
            self.losses.append(loss_item)
            self.accumulated += 1
        self.state, self.rsp_counter, _ = rsp_end_step(state, loss_nan, self.step, a, self.rsp_counter)
        if self.accumulated != a.grad_accumulation_batches:
            return None
        self.opt.step()
        rec = {"step": self.step, "loss": sum(self.losses), "lr": self.opt.learning_rate}
        self.accumulated = 0
        self.step += 1
        return rec
