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
    the hand-written transposed-read kernel (csrc/joint_wgrad.hip, `_joint_wgrad`) where its geometry fits (N, K
    multiples of 256), else the library: its transposed-A kernel tiles only the [N, K] output (8704 x 768 = 102 macro-tiles
    for 256 CUs) and runs at 0.85 PF/s; cut into 16 row chunks as ONE batched GEMM with fp32 outputs and summed, the same
    product takes 3.98 instead of 4.76 ms and the partial sums are fp32 instead of one bf16 rounding of the full sum.
    Both are deterministic: slices / chunks are added in a fixed order."""
    rows = dy2.shape[0]
    S = WGRAD_SPLIT
    if not dy2.is_cuda or dy2.dtype not in (torch.float16, torch.bfloat16):
        return torch.mm(dy2.t(), x2)
    if rows < 4096 * S:
        return torch.mm(dy2.t(), x2, out_dtype=torch.float32)
    if JOINT_WGRAD and dy2.is_contiguous() and x2.is_contiguous() and x2.dtype == dy2.dtype:
        out = _joint_wgrad(dy2, x2)
        if out is not None:
            return out
    per = rows // S
    main = per * S
    a = dy2[:main].view(S, per, dy2.shape[1])
    b = x2[:main].view(S, per, x2.shape[1])
    dw = torch.bmm(a.transpose(1, 2), b, out_dtype=torch.float32).sum(0)
    if main < rows:
        dw += torch.mm(dy2[main:].t(), x2[main:], out_dtype=torch.float32)
    return dw


# The joint projection on the hand-written MFMA GEMM (csrc/joint_gemm.hip): forward with the row log-sum-exp of the logits in
# its epilogue (the loss then skips its own pass over the 5.3 GB of logits), input gradient with the same kernel.  Default
# since round 4 (8-phase main loop, persistent workgroups: forward + LSE 4.0 ms against the library's 3.4 + 0.9, input
# gradient 2.8 against 2.95 at 304 000 x 768 x 8704).
# CAIMAN_JOINT_GEMM: "1" forward + input gradient (default), "fwd" forward only, "0" library GEMMs (F.linear / torch.mm).
JOINT_GEMM = __import__("os").environ.get("CAIMAN_JOINT_GEMM", "1")


def _joint_gemm(a2, w, bias, want_lse):
    """a2 [M, K] · w [N, K]^T (+ bias) -> (c [M, N], lse [M] f32 | None) through caiman_joint_fc_forward, or None when the
    shapes are outside the kernel's geometry."""
    from caiman_asr_amd import _lib

    M, K = a2.shape
    N = w.shape[0]
    tag = _lib.dtype_tag(a2.dtype)
    lib = _lib.lib()
    if not (a2.is_cuda and a2.dtype in (torch.float16, torch.bfloat16) and a2.is_contiguous() and w.is_contiguous()
            and w.dtype == a2.dtype and lib.caiman_joint_fc_supported(M, N, K, tag)):
        return None
    c = torch.empty((M, N), dtype=a2.dtype, device=a2.device)
    lse = ws = None
    if want_lse:
        lse = torch.empty((M,), dtype=torch.float32, device=a2.device)
        ws = torch.empty((lib.caiman_joint_fc_workspace_elems(M, N),), dtype=torch.float32, device=a2.device)
    # `nbytes` of the bracket carries the product's FLOP count (bench.py prices these kernels against the MFMA peak)
    with _lib.timed("joint_gemm_fwd" if want_lse else "joint_gemm_dx", 1, 2 * M * N * K):
        _lib.check(lib.caiman_joint_fc_forward(_lib.ptr(a2), _lib.ptr(w), None if bias is None else _lib.ptr(bias), _lib.ptr(c),
                                               None if lse is None else _lib.ptr(lse), None if ws is None else _lib.ptr(ws),
                                               M, N, K, tag, _lib.stream()))
    return c, lse


# The projection's weight gradient on the hand-written transposed-read GEMM (csrc/joint_wgrad.hip): 3.3-3.4 ms against the
# library's 3.9-4.0 at 304 000 x 8704 x 768 (tools/joint_gemm_bench.py).  CAIMAN_JOINT_WGRAD=0 restores the batched library call.
JOINT_WGRAD = __import__("os").environ.get("CAIMAN_JOINT_WGRAD", "1") != "0"


LIBRARY_TN_FLOPS = 0.8e15   # what the library's transposed-A GEMM reaches at the LSTM layers' shapes (tools/wgrad_tn_bench.py: 0.62-1.03)


def _wgrad_group_ok(dy3, x3, P, M, N, K):
    return (dy3.is_cuda and dy3.dtype in (torch.float16, torch.bfloat16) and x3.dtype == dy3.dtype
            and tuple(dy3.shape) == (P, M, N) and tuple(x3.shape) == (P, M, K)
            and dy3.stride(2) == 1 and dy3.stride(1) == N and x3.stride(2) == 1 and x3.stride(1) == K
            and (P == 1 or (dy3.stride(0) % 8 == 0 and x3.stride(0) % 8 == 0 and dy3.stride(0) >= 0 and x3.stride(0) >= 0))
            and dy3.data_ptr() % 16 == 0 and x3.data_ptr() % 16 == 0)


def wgrad_tn_estimate_us(M, N, K, P, dtype):
    """what the kernel's cost model expects `P` products [M, N]^T . [M, K] to take in one launch (us; < 0: not supported)"""
    from caiman_asr_amd import _lib

    return _lib.lib().caiman_wgrad_tn_estimate_us(M, N, K, P, _lib.dtype_tag(dtype))


def wgrad_tn(dy3, x3, only_if_faster=False, second=None, raw=False):
    """dy3 [P, M, N]^T . x3 [P, M, K] per p -> [P, N, K] fp32 through caiman_wgrad_tn (csrc/joint_wgrad.hip: slices of M into
    fp32 slabs, added in order, plus the library product of the few rows the slices do not cover), or None when the shapes or
    strides are outside the kernel (rows must be contiguous, the P operands a constant stride apart) -- or, with
    `only_if_faster`, when the kernel's own cost model expects the library to be quicker (short reductions whose tile
    count fills the last round of 256 workgroups badly).  `second` = (dy3b, x3b): a second strided group of products of the
    SAME shape in the same launch; the result then holds its products behind the first group's.  `raw`: the slabs themselves
    [P, slices, N, K] for a consumer that adds them up on its way (caiman_lstm_grad_deliver) -- only when the kernel has
    covered every row; otherwise the summed [P, N, K] as usual (tell them apart by the number of dimensions)."""
    import ctypes

    from caiman_asr_amd import _lib

    P, M, N = dy3.shape
    K = x3.shape[2]
    if not _wgrad_group_ok(dy3, x3, P, M, N, K):
        return None
    P2 = 0
    if second is not None:
        dyb, xb = second
        P2 = dyb.shape[0]
        if not _wgrad_group_ok(dyb, xb, P2, M, N, K) or dyb.dtype != dy3.dtype:
            return None
    sy, sx = (dy3.stride(0), x3.stride(0)) if P > 1 else (0, 0)
    lib, tag = _lib.lib(), _lib.dtype_tag(dy3.dtype)
    PT = P + P2
    if only_if_faster:
        est = lib.caiman_wgrad_tn_estimate_us(M, N, K, PT, tag)
        if est < 0 or est * 1e-6 > 2.0 * PT * M * N * K / LIBRARY_TN_FLOPS:
            return None
    per = ctypes.c_int64(0)
    slices = lib.caiman_wgrad_tn_plan(M, N, K, PT, tag, ctypes.byref(per))
    if slices <= 0:
        return None
    slabs = torch.empty((PT, slices, N, K), dtype=torch.float32, device=dy3.device)
    if P2:
        syb, sxb = (dyb.stride(0), xb.stride(0)) if P2 > 1 else (0, 0)
        _lib.check(lib.caiman_wgrad_tn2(_lib.ptr(dy3), sy, _lib.ptr(x3), sx, P, _lib.ptr(dyb), syb, _lib.ptr(xb), sxb, P2,
                                        _lib.ptr(slabs), M, N, K, slices, per.value, tag, _lib.stream()))
    else:
        _lib.check(lib.caiman_wgrad_tn(_lib.ptr(dy3), sy, _lib.ptr(x3), sx, _lib.ptr(slabs), P, M, N, K, slices, per.value, tag,
                                       _lib.stream()))
    done = slices * per.value
    if raw and (done == M or lib.caiman_wgrad_tn_covers_remainder(M, N, K, slices, per.value)):
        return slabs
    dw = slabs.sum(1) if slices > 1 else slabs[:, 0]
    if done < M and not lib.caiman_wgrad_tn_covers_remainder(M, N, K, slices, per.value):   # else: they rode in the last slice
        dw[:P] += torch.bmm(dy3[:, done:].transpose(1, 2), x3[:, done:], out_dtype=torch.float32)
        if P2:
            dw[P:] += torch.bmm(dyb[:, done:].transpose(1, 2), xb[:, done:], out_dtype=torch.float32)
    return dw


def _joint_wgrad(dy2, x2, raw=False):
    """The joint projection's instance: dy2 [M, N]^T . x2 [M, K] -> [N, K] fp32 (`raw`: possibly the slabs [slices, N, K],
    see wgrad_tn), or None (shape outside the kernel)."""
    if dy2.shape[0] < 512:
        return None
    from caiman_asr_amd import _lib

    with _lib.timed("joint_gemm_dw", 1, 2 * dy2.shape[0] * dy2.shape[1] * x2.shape[1]):
        out = wgrad_tn(dy2.unsqueeze(0), x2.unsqueeze(0), raw=raw)
    return None if out is None else out[0]


class _LinearTransposedBackward(torch.autograd.Function):
    """F.linear whose input gradient is computed against a [K, N] copy of the weight.  For the joint projection
    (rows x 768 x 8704, bf16) the library's `dY · Wᵀᵀ` (NT) kernel runs at 1.36 PF/s where the usual `dY · W` (NN)
    reaches 1.22 (tools/gemm_layout_bench.py); the 13 MB transposed copy is noise next to the 4 TFLOP GEMM."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias):
        from caiman_asr_amd.rnnt_ext.transducer.loss import clear_row_lse, offer_row_lse

        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        clear_row_lse()            # a projection that never reached a loss must not leave its normalisers behind
        if JOINT_GEMM != "0" and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16):
            x2 = x.reshape(-1, x.shape[-1])
            w16 = weight.to(x.dtype).contiguous()
            ctx.save_for_backward(x, w16)      # the backward pass transposes this 16-bit image instead of casting again
            out = _joint_gemm(x2 if x2.is_contiguous() else x2.contiguous(), w16,
                              None if bias is None else bias.to(x.dtype).contiguous(), want_lse=True)
            if out is not None:
                c, lse = out
                c = c.view(*x.shape[:-1], weight.shape[0])
                offer_row_lse(c, lse)      # the loss picks the normalisers up instead of reading the logits again
                return c
        else:
            ctx.save_for_backward(x, weight)
        return F.linear(x, weight, bias)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, w_saved = ctx.saved_tensors           # w_saved: the 16-bit image of the weight where the forward made one
        weight, _bias = ctx.params
        dy2 = dy.reshape(-1, dy.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = w_saved.to(dy2.dtype).t().contiguous()           # [K, N]
            out = _joint_gemm(dy2 if dy2.is_contiguous() else dy2.contiguous(), wt, None, want_lse=False) \
                if JOINT_GEMM == "1" else None
            dx = (out[0] if out is not None else torch.mm(dy2, wt.t())).view(*dy.shape[:-1], weight.shape[1])
        if ctx.needs_input_grad[1]:
            x2 = x.reshape(-1, x.shape[-1]).to(dy2.dtype)
            slabs = None
            if (DIRECT_GRADS and JOINT_WGRAD and weight.dtype == torch.float32 and weight.is_contiguous() and dy2.is_cuda
                    and dy2.dtype in (torch.float16, torch.bfloat16) and dy2.shape[0] >= 4096 * WGRAD_SPLIT
                    and dy2.is_contiguous() and x2.is_contiguous() and (weight.numel() % 4 == 0)):
                slabs = _joint_wgrad(dy2, x2, raw=True)
            if slabs is not None:
                # the slices' partial products go straight into `.grad` (summed in order), autograd gets None
                from caiman_asr_amd import _lib

                if weight.grad is None:
                    weight.grad = torch.zeros_like(weight)
                _lib.check(_lib.lib().caiman_slab_accumulate(_lib.ptr(slabs), slabs.shape[0] if slabs.dim() == 3 else 1,
                                                             weight.numel(), _lib.ptr(weight.grad), _lib.stream()))
                notify_grad_ready(weight)
            else:
                dw = _weight_gradient(dy2, x2).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            from caiman_asr_amd.rnnt_ext.transducer.loss import take_bias_gradient

            db = take_bias_gradient(dy)     # summed inside the loss backward kernel when dy comes straight from it
            db = (dy2.sum(0) if db is None else db.view(-1)).to(weight.dtype)
        return dx, dw, db


def linear_transposed_backward(x, weight, bias):
    return _LinearTransposedBackward.apply(x, weight, bias)


# Parameter gradients of the joint's input layers are accumulated by the GEMM that computes them (beta = 1 on the fp32
# `.grad`, autograd gets None) instead of a product, a widening copy and AccumulateGrad's add.  CAIMAN_DIRECT_GRADS=0: returned.
DIRECT_GRADS = __import__("os").environ.get("CAIMAN_DIRECT_GRADS", "1") != "0"


class _LinearF32Grads(torch.autograd.Function):
    """torch.nn.Linear under autocast, except that the weight and bias gradients leave the GEMM / the column sum as fp32
    (autocast's own backward computes them in the 16-bit type and widens afterwards: one more rounding of every element of
    the parameter gradient, up to 2^-8 of the tensor's range -- the reference does the same,
    training/caiman_asr_train/rnnt/model.py:409-439 under torch.autocast; here parameter gradients are never rounded).
    Used for joint_enc / joint_pred; the projection joint_fc has its own function above."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias):
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        # one GEMM with the bias in its epilogue, on the flattened CONTIGUOUS rows: handed a transposed view (the model passes
        # y.transpose(0, 1)), torch's linear falls back to matmul + a separate bias add -- a second rounding of every output
        # element in the 16-bit type (29 % of the elements of g one to eleven ulps off, tools/bf16_forward_probe.py).
        # The 16-bit images of the rows and of the weight are made once, here, and kept for the backward pass.
        dt = torch.get_autocast_dtype("cuda") if (x.is_cuda and torch.is_autocast_enabled("cuda")) else x.dtype
        x2 = x.reshape(-1, x.shape[-1]).to(dt)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w = weight.to(dt)
        ctx.save_for_backward(x2, w)
        ctx.x_dtype = x.dtype
        if bias is None:
            return torch.mm(x2, w.t()).view(*x.shape[:-1], weight.shape[0])
        return torch.addmm(bias.to(dt), x2, w.t()).view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        weight, bias = ctx.params
        dy2 = dy.reshape(-1, dy.shape[-1]).to(x2.dtype)
        low = dy2.is_cuda and dy2.dtype in (torch.float16, torch.bfloat16)
        direct = DIRECT_GRADS and low and weight.dtype == torch.float32
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(dy2, w).view(*dy.shape[:-1], w.shape[1]).to(ctx.x_dtype)
        if ctx.needs_input_grad[1]:
            if direct:
                if weight.grad is None:
                    weight.grad = torch.zeros_like(weight)
                torch.addmm(weight.grad, dy2.t(), x2, out_dtype=torch.float32, out=weight.grad)
                notify_grad_ready(weight)
            else:
                dw = (torch.mm(dy2.t(), x2, out_dtype=torch.float32) if low else torch.mm(dy2.t(), x2)).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            # (the column sum as a product dy^T . 1 with a one-column operand takes the library 11 ms of HOST time per call on
            # this stack, tools/torch_out_dtype_probe.py: the reduction stays a reduction)
            db = dy2.sum(0, dtype=torch.float32 if low else dy2.dtype).to(bias.dtype)
        return dx, dw, db


def linear_f32_grads(x, weight, bias):
    return _LinearF32Grads.apply(x, weight, bias)


class _EmbeddingDirectGrad(torch.autograd.Function):
    """torch.nn.functional.embedding whose weight gradient is added into `weight.grad` by one kernel with a fixed summation
    order (include/caiman_rnnt.h caiman_embedding_grad) instead of the library's sort-based embedding_dense_backward (90 us
    for the prediction network's 2 600 tokens) followed by AccumulateGrad's add."""

    @staticmethod
    def forward(ctx, idx, weight):
        ctx.save_for_backward(idx)
        ctx.weight = weight
        return F.embedding(idx, weight)

    @staticmethod
    def backward(ctx, dy):
        from caiman_asr_amd import _lib

        (idx,) = ctx.saved_tensors
        weight = ctx.weight
        if weight.grad is None:
            weight.grad = torch.zeros_like(weight)
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        tokens = idx.reshape(-1).long().contiguous()
        _lib.check(_lib.lib().caiman_embedding_grad(_lib.ptr(tokens), tokens.numel(), _lib.ptr(dy2), _lib.dtype_tag(dy2.dtype),
                                                    weight.shape[0], weight.shape[1], _lib.ptr(weight.grad), _lib.stream()))
        notify_grad_ready(weight)
        return None, None


def embedding(module, idx):
    """`module(idx)` for a torch.nn.Embedding; in training on the GPU with an fp32 table the gradient goes through
    _EmbeddingDirectGrad (plain lookups only: no padding index, no max-norm, dense gradients)."""
    w = module.weight
    if (DIRECT_GRADS and w.is_cuda and w.dtype == torch.float32 and w.requires_grad and torch.is_grad_enabled() and w.is_contiguous()
            and module.padding_idx is None and module.max_norm is None and not module.sparse and not module.scale_grad_by_freq):
        return _EmbeddingDirectGrad.apply(idx, w)
    return module(idx)
