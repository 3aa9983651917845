"""FusedLAMB on a flat fp32 arena (+ fused EMA and gradient zeroing) backed by csrc/optimizer.hip.

Constructor keywords follow apex.optimizers.FusedLAMB as the reference calls it
(training/caiman_asr_train/train_utils/build_optimizer.py:11-32).  On construction every
parameter is re-homed into ONE contiguous fp32 buffer (param.data and param.grad become views),
so the optimiser is three streaming passes over HBM and data-parallel training can all-reduce
contiguous slices of the gradient arena without bucket copies.
"""
from argparse import Namespace
from typing import Optional

import torch

from caiman_asr_amd import _lib

_ALIGN = 64          # elements; keeps every tensor 256-byte aligned in the arena
_CHUNK = 1 << 16     # elements per workgroup chunk


class FusedLAMB(torch.optim.Optimizer):
    # torch.amp.GradScaler.step(): hand over `grad_scale` / `found_inf` instead of unscaling tensor by tensor and
    # syncing on the host; step() divides the arena by the scale and caiman_lamb_step drops a non-finite step itself
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, bias_correction=True, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01,
                 amsgrad=False, adam_w_mode=True, grad_averaging=True, set_grad_none=True, max_grad_norm=1.0,
                 use_nvlamb=False, ema_decay: Optional[float] = None):
        if amsgrad:
            raise RuntimeError("FusedLAMB does not support the AMSGrad variant.")
        if not adam_w_mode or use_nvlamb:
            raise RuntimeError("only adam_w_mode=True, use_nvlamb=False (the reference's setting) is implemented")
        defaults = dict(lr=lr, bias_correction=bias_correction, betas=betas, eps=eps, weight_decay=weight_decay,
                        grad_averaging=grad_averaging, max_grad_norm=max_grad_norm)
        super().__init__(params, defaults)
        if len(self.param_groups) > 16:
            raise RuntimeError("at most 16 parameter groups")
        self.ema_decay = ema_decay
        self.set_grad_none = False  # grads are persistent views into the arena
        self._build_arena()

    # ---- arena ----------------------------------------------------------------------
    def _build_arena(self):
        plist, pgroup = [], []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if not p.requires_grad:
                    continue
                if p.dtype != torch.float32:
                    raise RuntimeError("FusedLAMB arena expects fp32 master parameters")
                if not p.is_cuda:
                    raise RuntimeError("FusedLAMB: parameters must be CUDA tensors (no CPU optimiser path)")
                plist.append(p)
                pgroup.append(gi)
        if not plist:
            raise RuntimeError("FusedLAMB: no trainable parameters")
        dev = plist[0].device
        offsets, total = [], 0
        for p in plist:
            offsets.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        chunk_start, chunk_len, chunk_tensor, first_chunk = [], [], [], []
        for ti, (p, off) in enumerate(zip(plist, offsets)):
            n = p.numel()
            view = self.flat_p[off:off + n].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_g[off:off + n].view(p.shape)
            first_chunk.append(len(chunk_start))
            for s in range(0, n, _CHUNK):
                chunk_start.append(off + s)
                chunk_len.append(min(_CHUNK, n - s))
                chunk_tensor.append(ti)
        first_chunk.append(len(chunk_start))
        self.flat_ema = self.flat_p.clone() if self.ema_decay is not None else None
        self._params, self._offsets = plist, offsets
        self._n_chunks, self._n_tensors = len(chunk_start), len(plist)
        self._chunk_start = torch.tensor(chunk_start, dtype=torch.int64, device=dev)
        self._chunk_len = torch.tensor(chunk_len, dtype=torch.int32, device=dev)
        self._chunk_tensor = torch.tensor(chunk_tensor, dtype=torch.int32, device=dev)
        self._first_chunk = torch.tensor(first_chunk, dtype=torch.int64, device=dev)
        self._tensor_group = torch.tensor(pgroup, dtype=torch.int32, device=dev)
        self._work = torch.zeros(8 + 2 * self._n_chunks + self._n_tensors, dtype=torch.float32, device=dev)
        self._step = torch.zeros(1, dtype=torch.int32, device=dev)

    @property
    def grad_norm(self) -> torch.Tensor:
        """Global gradient L2 norm of the last step() (device scalar, no sync)."""
        return self._work[0]

    @property
    def last_step_applied(self) -> torch.Tensor:
        return self._work[2]

    @property
    def last_step_dropped_for_handoff(self) -> torch.Tensor:
        """1 when the last step() was dropped because a weight-resident LSTM launch had timed out at a hand-off since
        the step before (device scalar, no sync; include/caiman_rnnt.h, caiman_lamb_step)."""
        return self._work[6]

    def zero_grad(self, set_to_none: bool = False):
        self.flat_g.zero_()

    def ema_tensors(self):
        """{parameter -> EMA view}; the EMA model is what the reference evaluates / exports."""
        assert self.flat_ema is not None
        return {p: self.flat_ema[o:o + p.numel()].view(p.shape) for p, o in zip(self._params, self._offsets)}

    # ---- checkpointing: the arenas are the state ------------------------------------------------------
    def state_dict(self):
        return {"flat_m": self.flat_m.detach().cpu(), "flat_v": self.flat_v.detach().cpu(),
                "flat_ema": None if self.flat_ema is None else self.flat_ema.detach().cpu(),
                "step": int(self._step.item()),
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
                "layout": [(tuple(p.shape), o) for p, o in zip(self._params, self._offsets)]}

    def load_state_dict(self, sd):
        layout = [(tuple(p.shape), o) for p, o in zip(self._params, self._offsets)]
        if [tuple(x) for x in map(lambda t: (tuple(t[0]), t[1]), sd["layout"])] != layout:
            raise RuntimeError("optimizer checkpoint does not match this model's parameter arena")
        self.flat_m.copy_(sd["flat_m"])
        self.flat_v.copy_(sd["flat_v"])
        if self.flat_ema is not None and sd.get("flat_ema") is not None:
            self.flat_ema.copy_(sd["flat_ema"])
        self._step.fill_(int(sd["step"]))
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update(saved)

    @torch.no_grad()
    def step(self, closure=None, inv_grad_scale: float = 1.0, zero_grad: bool = False):
        import ctypes

        # The decision to drop a step after a hand-off timeout is taken on the device (the host runs ahead of it);
        # the host only reports it, once: the failure word is host memory, reading it costs nothing.
        fails = int(_lib.lib().caiman_lstm_resident_failures())
        if fails > getattr(self, "_handoff_failures_reported", 0):
            import warnings

            warnings.warn(f"{fails} hand-off timeout(s) in the weight-resident LSTM kernels: the optimiser drops the "
                          "affected step(s) and the process stays on the per-timestep LSTM kernels "
                          "(caiman_lstm_resident_set_failures(0) re-admits the resident ones)")
        self._handoff_failures_reported = fails
        grad_scale = getattr(self, "grad_scale", None)   # set by GradScaler.step() for the duration of the call
        if grad_scale is not None:
            self.flat_g.div_(grad_scale.to(self.flat_g.device))
        g0 = self.param_groups[0]
        n_groups = len(self.param_groups)
        lrs = (ctypes.c_float * n_groups)(*[float(g["lr"]) for g in self.param_groups])
        wds = (ctypes.c_float * n_groups)(*[float(g["weight_decay"]) for g in self.param_groups])
        with _lib.timed("lamb"):
            _lib.check(_lib.lib().caiman_lamb_step(
                _lib.ptr(self.flat_p), _lib.ptr(self.flat_g), _lib.ptr(self.flat_m), _lib.ptr(self.flat_v),
                _lib.ptr(self.flat_ema) if self.flat_ema is not None else None, _lib.ptr(self._chunk_start),
                _lib.ptr(self._chunk_len), _lib.ptr(self._chunk_tensor), self._n_chunks, _lib.ptr(self._tensor_group),
                _lib.ptr(self._first_chunk), self._n_tensors, ctypes.cast(lrs, ctypes.c_void_p),
                ctypes.cast(wds, ctypes.c_void_p), n_groups, float(g0["betas"][0]), float(g0["betas"][1]),
                float(g0["eps"]), float(g0["max_grad_norm"] or 0.0),
                float(self.ema_decay if self.ema_decay is not None else 0.0), float(inv_grad_scale),
                int(bool(g0["bias_correction"])), int(bool(g0["grad_averaging"])), int(zero_grad),
                _lib.ptr(self._work), _lib.ptr(self._step), _lib.stream()))
        return None


def build_fused_lamb(args: Namespace, model, opt_eps: float) -> FusedLAMB:
    kw = {"params": model.param_groups(args.lr), "lr": args.lr, "weight_decay": args.weight_decay}
    return FusedLAMB(betas=(args.beta1, args.beta2), eps=opt_eps, max_grad_norm=args.clip_norm,
                     ema_decay=getattr(args, "ema", None) or None, **kw)


def build_optimizer(args: Namespace, model) -> FusedLAMB:
    """Top-level optimizer builder (eps = 1e-9, build_optimizer.py:27-32)."""
    return build_fused_lamb(args, model, 1e-9)


class OptimizerWrapper:
    """Optimiser + AMP scaling control of one training run (training/caiman_asr_train/train_utils/optimizer.py:11-57).

    * bf16 autocast or `--no_amp` (scaler None): `step()` is the optimiser's; the inf / NaN test the reference makes
      on the host (`np.isfinite(total_norm)`) is the device-side finite check of caiman_lamb_step.
    * fp16 autocast (a torch GradScaler): `scaler.step(optimizer)`; FusedLAMB declares `_step_supports_amp_scaling`,
      so the scaler hands it the loss scale and its found-inf flags without a host sync and the unscale is one
      division of the gradient arena.  `lower_bound` keeps the scale from collapsing: when `update()` has taken it
      below the bound, the next `update()` is told to set the bound instead (reference :38-47).
    """

    def __init__(self, args: Namespace, optimizer, scaler=None, lower_bound: Optional[float] = None, reducer=None):
        self.args, self.optimizer, self.scaler, self.lower_bound = args, optimizer, scaler, lower_bound
        self.reducer = reducer
        self.scale = None   # override handed to the next scaler.update()
        # a hand-off timeout drops the step on the device of the rank that saw it (caiman_lamb_step); the other ranks
        # only drop it too when the reducer carries the poison (distributed.py::guard_handoffs): pairing a reducer with
        # an optimiser therefore always arms the guard, otherwise the parameters would silently diverge across ranks
        if reducer is not None and hasattr(optimizer, "_work") and getattr(reducer, "_guard", None) is None:
            reducer.guard_handoffs(optimizer)

    def zero_grad(self) -> None:
        self.optimizer.zero_grad()

    def drop_window(self) -> None:
        """A NaN loss dropped the accumulation window (train.py:279-284): nothing of it may reach optimizer.step().
        Under data parallelism the reducer may hold marks (or, after a misuse, collectives in flight) of the dropped
        window: they are drained and forgotten before the next window's zero_grad touches the gradient arena."""
        if self.reducer is not None:
            self.reducer.reset()

    def step(self, total_norm: Optional[float] = None) -> None:
        """`total_norm` is accepted for signature parity; the finite test runs on the device."""
        if self.reducer is not None:
            self.reducer.finish()          # data parallel: gradients become the mean over ranks first
        if self.scaler is None:
            self.optimizer.step()
            return
        self.scaler.step(self.optimizer)
        self.scaler.update(self.scale)
        self.scale = None
        if self.lower_bound is not None and self.scaler.get_scale() < self.lower_bound:
            print("WARNING: Overriding the grad scaler")
            self.scale = self.lower_bound

    @property
    def learning_rate(self) -> float:
        return self.optimizer.param_groups[0]["lr"]
