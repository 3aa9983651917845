"""Hooks between the backward pass and the data-parallel gradient exchange.

* `register_grad_ready_callback`: LSTM stacks add their weight gradients straight into `param.grad` (views of the
  optimiser's flat fp32 arena) instead of returning them to autograd, and tell the listeners (the reducer of
  train_utils/distributed.py) per parameter, top layers first.
* `fence_collectives`: the weight-resident LSTM kernels and a collective's kernel never share the chip.
* `linear_transposed_backward`: the joint projection with its input gradient against a [K, N] weight copy.

Round 1 also carried an opt-in side-stream path (weight-gradient GEMMs and the prediction network on second
streams).  It was worth <= 1 % at batch 32 and its event graph could stall at batch >= 64: `flush_deferred()` made
the side stream wait on a gate event recorded on the main stream while `wait_all()` / the reducer made the main and
communication streams wait on the side stream, with thousands of per-timestep launches queued in between on the
same hardware queue.  An opt-in path that can stall is worse than none: it has been removed.
"""
import torch
import torch.nn.functional as F

_grad_ready_callbacks = []  # called as cb(param) right after a direct accumulation into param.grad has been queued


def register_grad_ready_callback(cb):
    _grad_ready_callbacks.append(cb)


def clear_grad_ready_callbacks():
    _grad_ready_callbacks.clear()


def notify_grad_ready(param):
    for cb in _grad_ready_callbacks:
        cb(param)


_comm_streams = []   # streams on which gradient collectives run (train_utils/distributed.py registers its own)


def register_comm_stream(stream):
    if stream is not None and stream not in _comm_streams:
        _comm_streams.append(stream)


def unregister_comm_stream(stream):
    if stream in _comm_streams:
        _comm_streams.remove(stream)


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


WGRAD_SPLIT = 16   # row chunks of the weight-gradient GEMM when the reduction dimension is long


def _weight_gradient(dy2, x2):
    """dW = dyᵀ · x for [rows, N] x [rows, K] with a very long `rows` (the joint projection: 300 000 lattice cells):
    the library's transposed-A kernel tiles only the [N, K] output (8704 x 768 = 102 macro-tiles for 256 CUs) and
    runs at 0.85 PF/s; cut into 16 row chunks as ONE batched GEMM with fp32 outputs and summed, the same product
    takes 3.98 instead of 4.76 ms (tools/dbg/dw_split2.py) and the partial sums are fp32 instead of one bf16 rounding
    of the full sum.  Deterministic: the chunks are added in a fixed order."""
    rows = dy2.shape[0]
    S = WGRAD_SPLIT
    if not dy2.is_cuda or dy2.dtype not in (torch.float16, torch.bfloat16) or rows < 4096 * S:
        return torch.mm(dy2.t(), x2)
    per = rows // S
    main = per * S
    a = dy2[:main].view(S, per, dy2.shape[1])
    b = x2[:main].view(S, per, x2.shape[1])
    dw = torch.bmm(a.transpose(1, 2), b, out_dtype=torch.float32).sum(0)
    if main < rows:
        dw += torch.mm(dy2[main:].t(), x2[main:], out_dtype=torch.float32)
    return dw


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
            dw = _weight_gradient(dy2, x.reshape(-1, x.shape[-1]).to(dy2.dtype)).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            from caiman_asr_amd.rnnt_ext.transducer.loss import take_bias_gradient

            db = take_bias_gradient(dy)     # summed inside the loss backward kernel when dy comes straight from it
            db = (dy2.sum(0) if db is None else db.view(-1)).to(weight.dtype)
        return dx, dw, db


def linear_transposed_backward(x, weight, bias):
    return _LinearTransposedBackward.apply(x, weight, bias)
