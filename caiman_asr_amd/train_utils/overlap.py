"""Side-stream execution of weight-gradient GEMMs.

In the backward pass only the INPUT gradients are on the critical path; the weight gradients are needed just
before the optimiser (or the all-reduce).  The LSTM backward that follows the joint's backward is a chain of
latency-bound step kernels that leaves the matrix cores idle, so the big `dW = dYᵀ·X` GEMMs are launched on a
second HIP stream and accumulate straight into the parameter's gradient (a view of the flat arena).
`wait_all()` must be called before the gradients are consumed (reducer.finish() / optimizer.step()).
"""
import os

import torch
import torch.nn.functional as F

_side = {}
_pending = False
_deferred = []   # closures that launch side-stream work; run by flush_deferred()
DEFER = os.environ.get("CAIMAN_DEFER", "1") != "0"   # hold the joint projection's weight-gradient GEMM back until the joint's own backward is queued
_grad_ready_callbacks = []  # called as cb(param) right after a side-stream accumulation has been queued


def register_grad_ready_callback(cb):
    _grad_ready_callbacks.append(cb)


def clear_grad_ready_callbacks():
    _grad_ready_callbacks.clear()


def side_streams():
    return list(_side.values())


_comm_streams = []   # streams on which gradient collectives run (train_utils/distributed.py registers its own)


def register_comm_stream(stream):
    if stream is not None and stream not in _comm_streams:
        _comm_streams.append(stream)


def fence_collectives():
    """Called at the start of an LSTM stack's backward pass: the weight-resident LSTM kernels want every CU of the
    chip (a resident workgroup fills its CU's register file), a collective's kernel holds some for as long as the
    slowest rank takes, and two kernels that are each placed by halves could wait for each other across ranks
    (DESIGN.md section 5).  So the stack's launches start only after the collectives queued so far are done; the
    collectives launched afterwards overlap ordinary kernels (weight-gradient GEMMs), as before."""
    if not _comm_streams:
        return
    cur = torch.cuda.current_stream()
    for s in _comm_streams:
        cur.wait_stream(s)


def side_stream(device) -> torch.cuda.Stream:
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _side:
        _side[key] = torch.cuda.Stream(device=device)
    return _side[key]


def flush_deferred():
    """Launch side-stream work that was held back.  A long GEMM launched on the side stream the moment its inputs
    exist fills every CU and starves the short kernel that comes next on the critical path (the joint's backward
    reduction went from 0.4 ms to 3.7 ms); so the producer only files the launch, and the next stage of the backward
    pass calls this right after queueing its own first kernel."""
    global _deferred
    if _deferred:
        todo, _deferred = _deferred, []
        # start behind what the caller has just queued, not next to it: the side streams wait for this point
        gate = torch.cuda.Event()
        gate.record(torch.cuda.current_stream())
        for s in _side.values():
            s.wait_event(gate)
        for launch in todo:
            launch()


def wait_all():
    """Make the current stream wait for every side-stream gradient GEMM issued so far."""
    global _pending
    flush_deferred()
    if _pending:
        for s in _side.values():
            torch.cuda.current_stream().wait_stream(s)
        _pending = False


def _accumulate(param, value):
    if param.grad is None:
        param.grad = torch.zeros_like(param)
    param.grad.add_(value.to(param.grad.dtype))
    for cb in _grad_ready_callbacks:
        cb(param)


class _LinearOverlapped(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.bias = bias
        return F.linear(x, weight, bias)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        global _pending
        x, weight = ctx.saved_tensors
        bias = ctx.bias
        dy = dy.contiguous()
        dx = torch.matmul(dy, weight.to(dy.dtype)) if ctx.needs_input_grad[0] else None
        main = torch.cuda.current_stream()
        side = side_stream(dy.device)
        ready = torch.cuda.Event()
        ready.record(main)          # dy and x are complete here, whatever the main stream does afterwards

        def launch():
            global _pending
            side.wait_event(ready)
            with torch.cuda.stream(side):
                dy2 = dy.reshape(-1, dy.shape[-1])
                x2 = x.reshape(-1, x.shape[-1]).to(dy.dtype)
                _accumulate(weight, torch.matmul(dy2.t(), x2))
                if bias is not None:
                    _accumulate(bias, dy2.sum(0))
            for t in (dy, x):
                t.record_stream(side)
            _pending = True

        if DEFER:
            _deferred.append(launch)
        else:
            launch()
        return dx, None, None


def linear_overlapped(x, weight, bias):
    """F.linear whose weight / bias gradients are produced on the side stream and added to `.grad` directly
    (autograd sees no gradient for them): call `wait_all()` before using the gradients."""
    return _LinearOverlapped.apply(x, weight, bias)


class _LinearTransposedBackward(torch.autograd.Function):
    """F.linear whose input gradient is computed against a [K, N] copy of the weight.  For the joint projection
    (rows x 768 x 8704, bf16) the library's `dY · Wᵀᵀ` (NT) kernel runs at 1.36 PF/s where the usual `dY · W` (NN)
    reaches 1.22 (tools/gemm_layout_bench.py); the 13 MB transposed copy is noise next to the 4 TFLOP GEMM."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = weight.to(dy2.dtype).t().contiguous()           # [K, N]
            dx = torch.mm(dy2, wt.t()).view(*dy.shape[:-1], weight.shape[1])
        if ctx.needs_input_grad[1]:
            dw = torch.mm(dy2.t(), x.reshape(-1, x.shape[-1]).to(dy2.dtype)).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy2.sum(0).to(weight.dtype)
        return dx, dw, db


def linear_transposed_backward(x, weight, bias):
    return _LinearTransposedBackward.apply(x, weight, bias)
