"""Data-parallel gradient exchange over RCCL (torch.distributed backend "nccl" on ROCm).

The reference wraps the model in torch DDP (training/caiman_asr_train/setup/train.py:190-196; three DDP
sub-modules under batch splitting, rnnt/sub_models.py:66-79): 25 MB buckets of copied gradients, mean
all-reduce during EVERY backward pass.  Here the gradients already live in ONE contiguous fp32 arena
(train_utils/optimizer.py), so a bucket is just a slice of that arena: no bucket copies, ONE exchange per
optimiser step, and the all-reduce of a slice is launched on a side stream as soon as the last gradient inside
it is final, overlapping the rest of the backward pass.  Arena order is parameter-group order (encoder,
prediction, joint_enc, joint_pred, joint_net) and backward produces gradients roughly in REVERSE arena order,
so buckets are cut from the tail.

Step protocol (one optimiser step may hold several backward passes: gradient accumulation, batch splitting):

    with reducer.no_sync():            # every backward pass except the last one that touches a parameter
        loss_i.backward()
    reducer.mark_ready(params)         # parameters whose last accumulation happened inside no_sync()
    loss_last.backward()               # hooks mark parameters final; a bucket goes out when all of its are
    reducer.finish()                   # launch what is left, wait, turn sums into means
    optimizer.step()

A gradient that is accumulated into a bucket whose collective is already in flight would be lost on the other
ranks (and race with the collective on this one): a hook that fires for such a parameter in a LATER backward pass
raises instead of letting that go silently.  (Within one pass a parameter may report twice -- the LSTM pipelines add
their weight gradients into `.grad` themselves and notify, and autograd's post-accumulate hook then fires for the same
parameter although it received no gradient from autograd -- which is harmless and ignored.)

xGMI is point-to-point (7 links x ~153 GB/s per GPU): few large messages beat many small ones,
so the default bucket is 64 MB (base model: 339 MB of fp32 gradients -> 6 collectives).
"""
import contextlib
from typing import Iterable, List

import torch
import torch.distributed as dist

from caiman_asr_amd.train_utils import overlap as _hooks


class FlatGradReducer:
    def __init__(self, params: List[torch.nn.Parameter], offsets: List[int], flat_grad: torch.Tensor,
                 process_group=None, bucket_bytes: int = 64 << 20, overlap: bool = True, measure_exposed: bool = False,
                 force_distributed: bool = False):
        """`force_distributed`: run the whole exchange (hooks, buckets, collectives, fences) even in a world of ONE rank,
        where a mean all-reduce is the identity -- the only way to execute the RCCL code path on a one-GPU box
        (tests/test_gpu_distributed.py); a normal run leaves it off and a world of one does nothing."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force_distributed and dist.is_initialized())
        self.launched_total = 0      # collectives launched since construction (tests, bench record)
        self.flat = flat_grad
        self.overlap = overlap and flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.overlap else None
        # cut buckets from the tail of the arena
        order = sorted(range(len(params)), key=lambda i: offsets[i], reverse=True)
        self.buckets = []  # (start, end, n_params)
        self.param_bucket = {}
        cur_end, cur_start, cur_n = flat_grad.numel(), flat_grad.numel(), 0
        limit = max(1, bucket_bytes // flat_grad.element_size())
        for i in order:
            cur_start = offsets[i]
            cur_n += 1
            self.param_bucket[id(params[i])] = len(self.buckets)
            if cur_end - cur_start >= limit:
                self.buckets.append((cur_start, cur_end, cur_n))
                cur_end, cur_n = cur_start, 0
        if cur_n:
            self.buckets.append((0, cur_end, cur_n))
        self._ready = [dict() for _ in self.buckets]    # per bucket: id of a parameter whose gradient is final -> pass
        self._pass = 0              # backward passes seen (advanced by an autograd end-of-pass callback)
        self._pass_cb_queued = False
        self._handles = []
        self._launched = [False] * len(self.buckets)
        self._syncing = True
        self._hooks = []
        # exposed time of the exchange: how long the compute stream stands still in finish() (device events, read by
        # exposed_ms(); the reference reports nothing comparable, SURVEY section 8(d).3 asks for it)
        self.measure_exposed = measure_exposed and self.overlap
        self._exposed_events = []
        # "nccl" (= RCCL) orders a collective against the stream wait() is called on; other backends (gloo: rehearsals,
        # CPU tests) complete on the host, so their handles are waited for in finish()
        self._stream_ordered = bool(dist.is_initialized() and self.overlap and dist.get_backend(process_group) == "nccl")
        self._guard = None       # (gradient element, failure count seen by the optimiser): see guard_handoffs()
        self._guard_bucket = None
        if self.active:
            for p in params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
            # gradients that LSTM stacks add straight into `.grad` bypass autograd's accumulation: they report here
            _hooks.register_grad_ready_callback(self._on_grad)
            _hooks.register_comm_stream(self.comm_stream)   # LSTM stacks fence against it (overlap.fence_collectives)

    def attach(self, model):
        """Make the reducer known to the train-step functions (train_utils/core.py, batch_splitting.py), which wrap
        their non-final backward passes in no_sync()."""
        getattr(model, "module", model).grad_reducer = self
        return self

    def guard_handoffs(self, optimizer):
        """A hand-off timeout of a weight-resident LSTM launch (csrc/lstm.hip) invalidates that rank's step without
        making anything non-finite; the optimiser drops such a step on the device (caiman_lamb_step), and every rank
        has to drop it with it.  With the guard on, the bucket at the head of the arena is only sent from finish(),
        behind a kernel that turns its first gradient into a NaN when the failure count has moved since the
        optimiser's last step (caiman_lstm_resident_poison): the sum carries the NaN to every rank."""
        self._guard = (self.flat[0:1], optimizer._work[5:6])
        self._guard_bucket = len(self.buckets) - 1
        assert self.buckets[self._guard_bucket][0] == 0
        return self

    # ---- per-step protocol -----------------------------------------------------------------------
    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside only accumulate (torch DDP's no_sync): nothing is marked final, nothing is sent."""
        prev, self._syncing = self._syncing, False
        try:
            yield
        finally:
            self._syncing = prev

    def mark_ready(self, params: Iterable[torch.nn.Parameter]):
        """The gradients of `params` are final although their last accumulation ran inside no_sync()."""
        for p in params:
            self._mark(p)

    def _end_of_pass(self):
        self._pass += 1
        self._pass_cb_queued = False

    def _on_grad(self, p):
        """post-accumulate-grad hook / notification from a layer pipeline: always inside a backward pass."""
        if self.active and not self._pass_cb_queued:
            self._pass_cb_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_pass)
        if self._syncing:
            self._mark(p)

    def _mark(self, p):
        b = self.param_bucket.get(id(p))
        if b is None or not self.active:
            return
        seen = self._ready[b].get(id(p))
        if seen is not None:
            if self._launched[b] and seen != self._pass:
                raise RuntimeError(
                    "FlatGradReducer: a gradient was accumulated into a bucket whose all-reduce is already in flight; "
                    "wrap every backward pass of an optimiser step except the last in reducer.no_sync() "
                    "(gradient accumulation, batch splitting)")
            return
        self._ready[b][id(p)] = self._pass
        if len(self._ready[b]) == self.buckets[b][2] and b != self._guard_bucket:
            self._launch(b)

    def _launch(self, b):
        if self._launched[b] or not self.active:
            return
        self._launched[b] = True
        self.launched_total += 1
        s, e, _ = self.buckets[b]
        chunk = self.flat[s:e]
        if self.overlap:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                h = dist.all_reduce(chunk, group=self.group, async_op=True)
                if self._stream_ordered:
                    # RCCL runs the collective on the process group's own stream; wait() does not block the host, it
                    # makes the CURRENT stream (comm_stream) wait for that stream.  From here on comm_stream stands
                    # for the collective, which is what overlap.fence_collectives() and finish() wait on: without it
                    # a weight-resident LSTM launch could be placed next to a collective still in flight.
                    h.wait()
                else:
                    self._handles.append(h)
        else:
            self._handles.append(dist.all_reduce(chunk, group=self.group, async_op=True))

    def finish(self, average: bool = True):
        """Launch whatever was not triggered by hooks (frozen / unused parameters, parameters last touched inside
        no_sync()), wait for all collectives and turn the sums into means.  Call after the last backward pass of
        the optimiser step, before optimizer.step()."""
        if self.active:
            if self._guard is not None and self.flat.is_cuda:
                from caiman_asr_amd import _lib

                _lib.check(_lib.lib().caiman_lstm_resident_poison(_lib.ptr(self._guard[0]), _lib.ptr(self._guard[1]),
                                                                  _lib.stream()))
            for b in range(len(self.buckets)):
                self._launch(b)
            for h in self._handles:
                h.wait()
            if self.overlap:
                cur = torch.cuda.current_stream()
                if self.measure_exposed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(cur)
                    cur.wait_stream(self.comm_stream)
                    e1.record(cur)
                    self._exposed_events.append((e0, e1))
                else:
                    cur.wait_stream(self.comm_stream)
            if average and self.world > 1:
                self.flat.mul_(1.0 / self.world)
        self._clear_step_state()

    def _clear_step_state(self):
        self._handles = []
        for r in self._ready:
            r.clear()
        self._launched = [False] * len(self.buckets)
        # the end-of-pass callback of a backward pass that raised never runs: do not let a stale "queued" flag stop the
        # pass counter (and with it the accumulated-into-an-in-flight-bucket check) for the rest of the run
        if self._pass_cb_queued:
            self._pass_cb_queued = False
            self._pass += 1

    def reset(self):
        """Abandon the current optimiser step (a NaN loss dropped the accumulation window, a backward pass raised):
        wait for every collective already in flight -- the gradient arena they write is about to be zeroed -- and
        forget which gradients were final.  The sums they produced are discarded with the window."""
        if self.active:
            for h in self._handles:
                h.wait()
            if self.overlap:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        self._clear_step_state()

    def exposed_ms(self, reset: bool = True) -> float:
        """Total time the compute stream waited for the collectives in finish() since the last reset (synchronises)."""
        if not self._exposed_events:
            return 0.0
        torch.cuda.synchronize()
        total = sum(a.elapsed_time(b) for a, b in self._exposed_events)
        if reset:
            self._exposed_events = []
        return total

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self._on_grad in _hooks._grad_ready_callbacks:
            _hooks._grad_ready_callbacks.remove(self._on_grad)
        _hooks.unregister_comm_stream(self.comm_stream)


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0, process_group=None):
    """DDP's constructor broadcast (rank 0 -> all), on the flat arena: one message."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.broadcast(flat_params, src=src, group=process_group)


def shard_utterances(n_items: int, rank: int, world: int):
    """Contiguous shard of a globally ordered (bucketed + shuffled) utterance list, as the
    reference's sampler / DALI reader do (training/caiman_asr_train/data/dali/sampler.py:225-262,
    pipeline.py:116-121,233-242)."""
    per = n_items // world
    return range(rank * per, (rank + 1) * per)
