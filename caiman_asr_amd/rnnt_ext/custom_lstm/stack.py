"""Layer-pipelined ("wavefront") execution of a whole LSTM stack on the gfx950 kernels.

The reference runs the layers of a stack one after the other, each as T dependent launches
(training/lib/src/rnnt_ext/custom_lstm/lstm.py:364-391 -> training/lib/csrc/lstm.cu:259-271), i.e. L*T
kernel boundaries.  On MI355X a boundary costs ~1.5 us of GPU time and the step kernel is latency-bound, so
the stack is restructured as a software pipeline over CHUNKS of timesteps:

    tick tau:  layer l advances chunk (tau - l)   for every l with 0 <= tau - l < n_chunks

All layers active in a tick are advanced by ONE launch per timestep (csrc/lstm.hip, blockIdx.z = layer),
which needs T + (L-1)*chunk launches instead of L*T.  Between ticks the input GEMM of each layer's next
chunk (x·Wᵀ + b on the chunk the layer below has just produced) is a plain library GEMM.  The backward
pass is the mirror image (top layer first, time reversed); weight gradients are batched GEMMs at the end.

Numerics are those of the per-layer path (same kernels, same GEMM operands); only the schedule differs.
"""
import ctypes

import torch

from caiman_asr_amd import _lib
from caiman_asr_amd.rnnt_ext.cuda.lstm import _step_bytes

CHUNK = int(__import__("os").environ.get("CAIMAN_LSTM_CHUNK", "32"))  # timesteps per pipeline chunk, shallow stacks
CHUNK_DEEP = int(__import__("os").environ.get("CAIMAN_LSTM_CHUNK_DEEP", "32"))  # stacks of >= 4 layers


CHUNK_SHORT = int(__import__("os").environ.get("CAIMAN_LSTM_CHUNK_SHORT", "8"))  # sequences of <= 128 steps


def _chunk(L, T=None):
    """A short sequence (the prediction network sees ~60 tokens) pays the (L-1)-chunk pipeline fill relatively
    more: smaller chunks there (measured 38.3 -> 38.1 ms per step), 32 otherwise."""
    if T is not None and T <= 128:
        return CHUNK_SHORT
    return CHUNK_DEEP if L >= 4 else CHUNK
IMAGES = int(__import__("os").environ.get("CAIMAN_LSTM_IMAGES", "1")) != 0   # csrc/lstm_images.hip: one launch for all weight images
# parameter gradients are ADDED into `param.grad` by one kernel per layer (un-permute + accumulate; autograd gets None), as
# in encoder_pipe.py; 0: returned to autograd after an un-permute copy each
EARLY_WGRAD = int(__import__("os").environ.get("CAIMAN_EARLY_WGRAD", "1")) != 0
INTERLEAVED = 1  # gate layout used INSIDE the pipeline: [.., H, 4] (see include/caiman_rnnt.h)
RINGS_ZEROED = 2  # caiman_lstm_prepare(gate_layout | RINGS_ZEROED): the caller has cleared ring / dC itself (one memset for all layers)


def _perm_rows(w, H):
    """rows [gate][unit] -> [unit][gate] (works for [4H, K] matrices and [4H] vectors)."""
    return w.reshape(4, H, *w.shape[1:]).transpose(0, 1).reshape(w.shape)


def _perm_cast_t(w, H, dt):
    """[4H, K] rows [gate][unit] -> [K, 4H] columns [unit][gate], cast, in one copy kernel."""
    K = w.shape[1]
    out = torch.empty((K, w.shape[0]), dtype=dt, device=w.device)
    out.view(K, H, 4).copy_(w.view(4, H, K).permute(2, 1, 0))
    return out


def _perm_cast(w, H, dt):
    """_perm_rows and a cast in one copy kernel."""
    out = torch.empty(w.shape, dtype=dt, device=w.device)
    out.view(H, 4, *w.shape[1:]).copy_(w.view(4, H, *w.shape[1:]).transpose(0, 1))
    return out


def _unperm_rows(w, H):
    """inverse of _perm_rows."""
    return w.reshape(H, 4, *w.shape[1:]).transpose(0, 1).reshape(w.shape)


def eligible(x: torch.Tensor, hidden_size: int, num_layers: int, gate_dtype) -> bool:
    return (x.is_cuda and num_layers >= 2 and num_layers <= 8 and hidden_size % 32 == 0
            and gate_dtype in (torch.float16, torch.bfloat16))


class _Scratch:
    """Per-(device, stream) cache of tiled weights / rings so successive steps reuse the allocations."""
    cache = {}

    @classmethod
    def get(cls, key, numel, dtype, device):
        k = (key, device, dtype, torch.cuda.current_stream(device).cuda_stream)
        t = cls.cache.get(k)
        if t is None or t.numel() < numel:
            t = torch.empty(numel, dtype=dtype, device=device)
            cls.cache[k] = t
        return t[:numel]


def _pad32(b):
    return (b + 31) // 32 * 32


class StackFunction(torch.autograd.Function):
    """forward(x [T,B,I], h0 [L,B,H], c0 [L,B,H], hard, p_drop, training, W_0, R_0, bW_0, bR_0, W_1, ...)
    -> (y_top [T,B,H], all_h [L,T,B,H], all_c [L,T,B,H]).  Gradients flow to x and the parameters.

    Inter-layer dropout is fused into the step kernels (counter-hash mask, no mask tensors): the forward epilogue of
    layer l also writes the masked copy that layer l+1 multiplies with W_ih, the backward epilogue of layer l masks
    the gradient arriving from layer l+1."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, h0, c0, hard, p_drop, training, *params):
        L = len(params) // 4
        Ws, Rs, bWs, bRs = params[0::4], params[1::4], params[2::4], params[3::4]
        T, B, _ = x.shape
        dev = x.device
        lib = _lib.lib()
        H = Rs[0].shape[1]
        # an output nobody differentiates (all_h when the states are only carried, y_top never) arrives as None in backward,
        # not as a tensor of zeros: the general path there (explicit dropout factors, one add per chunk) is for real gradients
        ctx.set_materialize_grads(False)
        # the pipeline keeps gates / dG unit-major ([.., H, 4]): permute the ROWS of W_ih and of the biases once
        # per call (R keeps its layout: the tiling kernels absorb the permutation)
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else x.dtype
        tag = _lib.dtype_tag(dt)
        bp = _pad32(B)
        wt = _Scratch.get("fw", L * 4 * H * H, dt, dev).view(L, -1)
        # all 16-bit images of the fp32 parameters in one launch (csrc/lstm_images.hip); the backward fragment images are
        # built here too and kept for the backward pass in place of a plain 16-bit copy of R
        fused_img = (IMAGES and dt in (torch.float16, torch.bfloat16) and L <= 16 and
                     all(p_.dtype == torch.float32 and p_.is_contiguous() and p_.is_cuda for p_ in params) and
                     all(W.shape[1] % 4 == 0 and W.data_ptr() % 16 == 0 for W in Ws) and all(R.data_ptr() % 16 == 0 for R in Rs))
        if fused_img:
            Wp = [torch.empty((W.shape[1], 4 * H), dtype=dt, device=dev) for W in Ws]
            bias = [torch.empty(4 * H, dtype=dt, device=dev) for _ in range(L)]
            Rp = [torch.empty(4 * H * H, dtype=dt, device=dev) for _ in range(L)]
            imgs = (_lib.LstmImages * L)(*[
                _lib.LstmImages(Ws[l].data_ptr(), Rs[l].data_ptr(), bWs[l].data_ptr(), bRs[l].data_ptr(), Wp[l].data_ptr(), None,
                                bias[l].data_ptr(), wt[l].data_ptr(), Rp[l].data_ptr(), H, Ws[l].shape[1]) for l in range(L)])
            _lib.check(lib.caiman_lstm_weight_images(ctypes.cast(imgs, ctypes.c_void_p), L, tag, _lib.stream()))
        else:
            Rp = [R.to(dt).contiguous() for R in Rs]
            Wp = [_perm_cast_t(W, H, dt) for W in Ws]    # [K, 4H]: NN forward, NT backward (tools/lstm_gemm_layout_bench.py)
            bias = [_perm_cast(bWs[l] + bRs[l], H, dt) for l in range(L)]
        G = torch.empty((L, T, B, 4 * H), dtype=dt, device=dev)
        torch.addmm(bias[0], x.flatten(0, 1).to(dt), Wp[0], out=G[0].view(T * B, 4 * H))
        Y = torch.empty((L, T + 1, B, H), dtype=dt, device=dev)
        Cs = torch.empty((L, T + 1, B, H), dtype=dt, device=dev)
        # initial state into row 0; None (no carried state): zeros without a tensor of zeros
        Y[:, 0].zero_() if h0 is None else Y[:, 0].copy_(h0)
        Cs[:, 0].zero_() if c0 is None else Cs[:, 0].copy_(c0)
        drop = float(p_drop) if (training and p_drop > 0.0 and L > 1) else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if drop > 0.0 else 0
        YM = torch.empty((L - 1, T, B, H), dtype=dt, device=dev) if drop > 0.0 else None  # masked outputs
        ring = _Scratch.get("fr", L * 2 * bp * H, dt, dev).view(L, -1)
        ring.zero_()     # the h rings of all layers in one memset
        st = _lib.stream()
        for l in range(L):
            if fused_img and h0 is None:
                continue     # nothing to prepare: the weight images exist and a zero initial state is what the cleared ring holds
            _lib.check(lib.caiman_lstm_prepare(None if fused_img else _lib.ptr(Rp[l]), _lib.ptr(Y[l, 0]), _lib.ptr(wt[l]),
                                                _lib.ptr(ring[l]), None, B, H, tag, 0, INTERLEAVED | RINGS_ZEROED, st))
        CH = _chunk(L, T)
        n_ch = (T + CH - 1) // CH
        sb = _step_bytes(B, H, G.element_size(), False)
        row = B * H
        for tau in range(n_ch + L - 1):
            slots = []
            for l in range(L):
                k = tau - l
                if k < 0 or k >= n_ch:
                    continue
                t0, n = k * CH, min(CH, T - k * CH)
                if l >= 1:  # input GEMM of this layer's chunk on what layer l-1 produced last tick
                    src = YM[l - 1, t0:t0 + n] if drop > 0.0 else Y[l - 1, 1 + t0:1 + t0 + n]
                    torch.addmm(bias[l], src.reshape(n * B, H), Wp[l], out=G[l, t0:t0 + n].view(n * B, 4 * H))
                masked = drop > 0.0 and l < L - 1
                slots.append(_lib.FwdSlot(wt[l].data_ptr(), G[l, t0].data_ptr(), Cs[l, t0].data_ptr(),
                                          Y[l, t0].data_ptr(), ring[l].data_ptr(), t0 & 1, n,
                                          YM[l, t0].data_ptr() if masked else None, (l * T + t0) * row,
                                          drop if masked else 0.0, 0))
            arr = (_lib.FwdSlot * len(slots))(*slots)
            n_launch = max(s_.nsteps for s_ in slots)
            with _lib.timed("lstm_fwd", n_launch, sb * sum(s_.nsteps for s_ in slots)) as tm:
                r0 = lib.caiman_lstm_resident_launches() if tm.start is not None else 0
                _lib.check(lib.caiman_lstm_wave_fwd(ctypes.cast(arr, ctypes.c_void_p), len(slots), n_launch, B, H,
                                                    tag, int(hard), INTERLEAVED, seed, st))
                if tm.start is not None and lib.caiman_lstm_resident_launches() != r0:   # weights read once per slot
                    tm.units = 1
                    tm.nbytes -= sum(s_.nsteps - 1 for s_ in slots) * 4 * H * H * G.element_size()
        saved = [x, G, Y, Cs, *Wp, *Rp]
        if YM is not None:
            saved.append(YM)
        ctx.save_for_backward(*saved)
        ctx.meta = (L, T, B, H, hard, drop, seed, x.requires_grad, fused_img)
        ctx.params = params
        y_top = Y[L - 1, 1:]
        all_h, all_c = Y[:, 1:], Cs[:, 1:]
        ctx.mark_non_differentiable(all_c)
        return y_top, all_h, all_c

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, d_top, d_allh, _d_allc):
        L, T, B, H, hard, drop, seed, need_dx, fused_img = ctx.meta
        from caiman_asr_amd.train_utils import overlap

        overlap.fence_collectives()
        saved = ctx.saved_tensors
        x, G, Y, Cs = saved[:4]
        Wp, Rp = saved[4:4 + L], saved[4 + L:4 + 2 * L]
        YM = saved[4 + 2 * L] if drop > 0.0 else None
        dev, dt = G.device, G.dtype
        tag = _lib.dtype_tag(dt)
        lib = _lib.lib()
        st = _lib.stream()
        row = B * H
        dG = torch.empty_like(G)
        # delta[l] = gradient w.r.t. the (masked) output sequence of layer l.  Top layer: what autograd hands us;
        # lower layers: dG_{l+1}·W_{l+1}, written chunk by chunk by the library GEMM; the dropout factor is applied
        # inside the step kernel.  A direct gradient on all_h (random state passing never produces one, states are
        # detached) takes the general path: materialise the factors and fold everything into delta.
        fused_mask = d_allh is None
        if d_top is None:
            d_top = torch.zeros((T, B, H), dtype=dt, device=dev)
        d_top = d_top.to(dt)
        if d_top.stride(2) != 1:
            d_top = d_top.contiguous()
        delta_low = torch.empty((L - 1, T, B, H), dtype=dt, device=dev) if L > 1 else None
        extra = None
        if not fused_mask:
            extra = d_allh.to(dt)
            d_top = d_top + extra[L - 1]
        bp = _pad32(B)
        wt = list(Rp) if fused_img else _Scratch.get("bw", L * 4 * H * H, dt, dev).view(L, -1)   # fused: the saved images
        # dG rings (16-bit) and dC carries (fp32) of all layers in one byte buffer: one memset instead of two per layer
        ring_b, dc_b = L * 2 * bp * 4 * H * G.element_size(), L * B * H * 4
        ring_pad = (ring_b + 15) // 16 * 16
        zero_all = _Scratch.get("bz_all", ring_pad + dc_b, torch.uint8, dev)
        zero_all.zero_()
        ring = zero_all[:ring_b].view(dt).view(L, -1)
        dC = zero_all[ring_pad:ring_pad + dc_b].view(torch.float32).view(L, -1)
        # bias gradients from the backward kernels (BwdSlot.dbias) where the weight-resident kernels run; otherwise one
        # reduction per layer at the end (cheaper than the per-timestep path's extra launch per call)
        fused_db = bool(lib.caiman_lstm_resident_would_run(B, H, min(8, L)))
        dbias = torch.zeros((L, 4 * H), dtype=torch.float32, device=dev) if fused_db else None
        for l in range(L):
            if fused_img:
                continue     # images saved by the forward pass, rings and carries cleared above: nothing left to prepare
            _lib.check(lib.caiman_lstm_prepare(_lib.ptr(Rp[l]), None, _lib.ptr(wt[l]), _lib.ptr(ring[l]),
                                                _lib.ptr(dC[l]), B, H, tag, 1, INTERLEAVED | RINGS_ZEROED, st))
        CH = _chunk(L, T)
        n_ch = (T + CH - 1) // CH
        sb = _step_bytes(B, H, G.element_size(), True)
        for tau in range(n_ch + L - 1):
            slots = []
            for l in range(L - 1, -1, -1):
                j = tau - (L - 1 - l)  # how many chunks this layer has already finished
                if j < 0 or j >= n_ch:
                    continue
                k = n_ch - 1 - j
                t0, n = k * CH, min(CH, T - k * CH)
                thi = t0 + n - 1
                p_slot = 0.0
                if l < L - 1:  # gradient from the layer above for this chunk: dX = dG_{l+1} @ W_{l+1}
                    out = delta_low[l, t0:t0 + n].view(n * B, H)
                    torch.matmul(dG[l + 1, t0:t0 + n].view(n * B, 4 * H), Wp[l + 1].t(), out=out)
                    if drop > 0.0:
                        if fused_mask:
                            p_slot = drop
                        else:
                            m = torch.empty((n, B, H), dtype=dt, device=dev)
                            _lib.check(lib.caiman_lstm_dropout_mask(_lib.ptr(m), m.numel(), seed, (l * T + t0) * row,
                                                                    drop, tag, st))
                            delta_low[l, t0:t0 + n].mul_(m)
                    if not fused_mask:
                        delta_low[l, t0:t0 + n] += extra[l, t0:t0 + n]
                    d = delta_low[l]
                else:
                    d = d_top
                slots.append(_lib.BwdSlot(wt[l].data_ptr(), G[l, thi].data_ptr(), Cs[l, thi].data_ptr(),
                                          d[thi].data_ptr(), d.stride(0), d.stride(1), dG[l, thi].data_ptr(),
                                          ring[l].data_ptr(), dC[l].data_ptr(), thi & 1, n, int(thi < T - 1),
                                          p_slot, (l * T + thi) * row, 0, 0, dbias[l].data_ptr() if fused_db else None))
            arr = (_lib.BwdSlot * len(slots))(*slots)
            n_launch = max(s_.nsteps for s_ in slots)
            with _lib.timed("lstm_bwd", n_launch, sb * sum(s_.nsteps for s_ in slots)) as tm:
                r0 = lib.caiman_lstm_resident_launches() if tm.start is not None else 0
                _lib.check(lib.caiman_lstm_wave_bwd(ctypes.cast(arr, ctypes.c_void_p), len(slots), n_launch, B, H,
                                                    tag, int(hard), INTERLEAVED, seed, st))
                if tm.start is not None and lib.caiman_lstm_resident_launches() != r0:   # weights read once per slot
                    tm.units = 1
                    tm.nbytes -= sum(s_.nsteps - 1 for s_ in slots) * 4 * H * H * G.element_size()

        def weight_grads(l, unperm=True):
            """[dW, dR, db, db] of layer l; unperm=False: rows left in the pipeline's [unit][gate] order, contiguous"""
            dg = dG[l].view(T * B, 4 * H)
            if l == 0:
                xin = x.detach().flatten(0, 1).to(dt)
            else:
                xin = (YM[l - 1] if drop > 0.0 else Y[l - 1, 1:]).reshape(T * B, H)
            # parameter gradients leave as fp32 products (a 16-bit library output would round every element once more:
            # up to 4e-3 of the tensor's range, profiles/r04_bf16_residual.md)
            f32 = dict(out_dtype=torch.float32) if dg.dtype in (torch.float16, torch.bfloat16) else {}
            fix = (lambda g_: _unperm_rows(g_, H)) if unperm else (lambda g_: g_)
            dB = fix(dbias[l] if fused_db else dg.sum(0, dtype=torch.float32))
            return [fix(torch.mm(dg.t(), xin, **f32)), fix(torch.mm(dg.t(), Y[l, :-1].reshape(T * B, H), **f32)), dB, dB]

        dX = torch.matmul(dG[0].view(T * B, 4 * H), Wp[0].t()).view(T, B, -1) if need_dx else None
        grads, pending = [], []

        def flush():
            """up to two layers (8 parameters) per launch: un-permute the gate rows and add into `.grad`"""
            if not pending:
                return
            items = (_lib.GradItem * len(pending))(*[
                _lib.GradItem(g_.data_ptr(), p_.grad.data_ptr(), H, p_.shape[1] if p_.dim() == 2 else 1,
                              int(g_.dtype == torch.float32), 0) for p_, g_ in pending])
            _lib.check(lib.caiman_lstm_grad_deliver(ctypes.cast(items, ctypes.c_void_p), len(pending), tag, st))
            for p_, _ in pending:
                overlap.notify_grad_ready(p_)
            pending.clear()

        for l in range(L):
            ps = ctx.params[4 * l:4 * l + 4]
            if (EARLY_WGRAD and IMAGES and dt in (torch.float16, torch.bfloat16)
                    and all(p_.requires_grad and p_.dtype == torch.float32 and p_.is_contiguous() for p_ in ps)):
                for p_ in ps:
                    if p_.grad is None:
                        p_.grad = torch.zeros_like(p_)
                pending += list(zip(ps, weight_grads(l, unperm=False)))   # the sources stay alive until the launch
                if len(pending) == 8:
                    flush()
                grads += [None] * 4
            else:
                grads += weight_grads(l)
        flush()
        return (dX, None, None, None, None, None, *grads)
