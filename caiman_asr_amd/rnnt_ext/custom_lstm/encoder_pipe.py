"""One layer pipeline across the whole encoder: pre_rnn -> StackTime -> post_rnn.

`stack.py` pipelines the layers of ONE stack; the encoder is two stacks with a time reduction between them
(training/caiman_asr_train/rnnt/model.py:314-342), so run stack after stack it pays the pipeline fill and drain
twice: (nA + La - 1) + (nB + Lb - 1) ticks of 32 dependent launches.  Here the first post layer starts on chunk j
as soon as the top pre layer has produced the 2·32 frames it stacks:

    pre  layer l takes chunk k at tick            k + l
    post layer m takes chunk j at tick   min(2j + 1, nA - 1) + La + m        (factor 2; f in general)

which ends after about nA + La + Lb ticks (base encoder, 430-frame batch: 21 ticks instead of 27).  Kernels, GEMM
operands and numerics are those of stack.py; the dropout between the two stacks (the final dropout of pre_rnn,
rnn.py:200-206) is applied by the step kernel like an inter-layer dropout.  The backward pass walks the same
schedule in reverse.
"""
import ctypes

import torch

from caiman_asr_amd import _lib
from caiman_asr_amd.rnnt_ext.cuda.lstm import _step_bytes
from caiman_asr_amd.rnnt_ext.custom_lstm import stack
from caiman_asr_amd.rnnt_ext.custom_lstm.stack import INTERLEAVED, _pad32, _perm_rows, _Scratch, _unperm_rows

CH = int(__import__("os").environ.get("CAIMAN_ENC_PIPE_CHUNK", "32"))   # timesteps per pipeline chunk


def eligible(x, hidden, La, Lb, gate_dtype, factor):
    return (x.is_cuda and La >= 1 and Lb >= 1 and La + Lb <= 8 and hidden % 32 == 0 and factor >= 1
            and gate_dtype in (torch.float16, torch.bfloat16))


def _schedule(nA, nB, La, Lb, f):
    """-> list of ticks, each a list of (layer, chunk)."""
    ticks = {}
    for l in range(La):
        for k in range(nA):
            ticks.setdefault(k + l, []).append((l, k))
    for m in range(Lb):
        for j in range(nB):
            ready = min(f * j + f - 1, nA - 1) + La      # the last pre chunk it stacks is done at the end of tick (c + La - 1)
            ticks.setdefault(ready + m, []).append((La + m, j))
    return [sorted(ticks[t]) for t in sorted(ticks)]


def _stacked(src, t0, n, f, B, H):
    """rows of StackTime for post steps [t0, t0+n): src [T1p, B, H] -> [n*B, f*H]."""
    return src[f * t0:f * (t0 + n)].view(n, f, B, H).transpose(1, 2).reshape(n * B, f * H)


class EncoderPipeFunction(torch.autograd.Function):
    """forward(x [T1,B,I], h0a, c0a [La,B,H], h0b, c0b [Lb,B,H], hard, p_drop, training, factor, La, *params)
    -> (y_top [T2,B,H], all_h_a [La,T1,B,H], all_c_a, all_h_b [Lb,T2,B,H], all_c_b); params = (W, R, bW, bR) per layer,
    pre layers first.  Gradients flow to x and the parameters (not to the all_* outputs: states are detached)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, h0a, c0a, h0b, c0b, hard, p_drop, training, factor, La, *params):
        L = len(params) // 4
        Lb, f = L - La, int(factor)
        Ws, Rs, bWs, bRs = params[0::4], params[1::4], params[2::4], params[3::4]
        T1, B, _ = x.shape
        T2 = (T1 + f - 1) // f
        T1p = T2 * f
        dev, lib = x.device, _lib.lib()
        H = Rs[0].shape[1]
        Tl = [T1] * La + [T2] * Lb
        g0 = torch.addmm(_perm_rows(bWs[0] + bRs[0], H), x.flatten(0, 1), _perm_rows(Ws[0], H).t())
        dt = g0.dtype
        tag = _lib.dtype_tag(dt)
        Ga = torch.empty((La, T1, B, 4 * H), dtype=dt, device=dev)
        Gb = torch.empty((Lb, T2, B, 4 * H), dtype=dt, device=dev)
        Ga[0].copy_(g0.view(T1, B, 4 * H))
        del g0
        # pre outputs carry f-1 zero frames at the end so that the last stacked frame is zero padded (StackTime)
        Ya = torch.zeros((La, T1p + 1, B, H), dtype=dt, device=dev) if T1p != T1 else torch.empty((La, T1 + 1, B, H), dtype=dt, device=dev)
        Ca = torch.empty((La, T1 + 1, B, H), dtype=dt, device=dev)
        Yb = torch.empty((Lb, T2 + 1, B, H), dtype=dt, device=dev)
        Cb = torch.empty((Lb, T2 + 1, B, H), dtype=dt, device=dev)
        Ya[:, 0].copy_(h0a)
        Ca[:, 0].copy_(c0a)
        Yb[:, 0].copy_(h0b)
        Cb[:, 0].copy_(c0b)
        G = [Ga[l] for l in range(La)] + [Gb[m] for m in range(Lb)]
        Y = [Ya[l] for l in range(La)] + [Yb[m] for m in range(Lb)]
        C = [Ca[l] for l in range(La)] + [Cb[m] for m in range(Lb)]
        Rp = [R.to(dt).contiguous() for R in Rs]
        Wp = [_perm_rows(W, H).to(dt) for W in Ws]
        bias = [_perm_rows(bWs[l] + bRs[l], H).to(dt) for l in range(L)]
        drop = float(p_drop) if (training and p_drop > 0.0) else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if drop > 0.0 else 0
        # masked outputs of every layer but the top one (the boundary layer La-1 included)
        YMa = (torch.zeros((La, T1p, B, H), dtype=dt, device=dev) if T1p != T1 else torch.empty((La, T1, B, H), dtype=dt, device=dev)) \
            if drop > 0.0 else None
        YMb = torch.empty((max(Lb - 1, 1), T2, B, H), dtype=dt, device=dev) if drop > 0.0 else None
        YM = ([YMa[l] for l in range(La)] + [YMb[m] for m in range(Lb - 1)] + [None]) if drop > 0.0 else [None] * L
        row = B * H
        base = [0] * L   # dropout counter base of each layer: disjoint ranges
        for l in range(1, L):
            base[l] = base[l - 1] + Tl[l - 1] * row
        bp = _pad32(B)
        wt = _Scratch.get("ep_fw", L * 4 * H * H, dt, dev).view(L, -1)
        ring = _Scratch.get("ep_fr", L * 2 * bp * H, dt, dev).view(L, -1)
        st = _lib.stream()
        for l in range(L):
            _lib.check(lib.caiman_lstm_prepare(_lib.ptr(Rp[l]), _lib.ptr(Y[l][0]), _lib.ptr(wt[l]), _lib.ptr(ring[l]),
                                                None, B, H, tag, 0, INTERLEAVED, st))
        nA, nB = (T1 + CH - 1) // CH, (T2 + CH - 1) // CH
        sb = _step_bytes(B, H, Ga.element_size(), False)
        sched = _schedule(nA, nB, La, Lb, f)
        for tick in sched:
            slots = []
            for l, k in tick:
                t0, n = k * CH, min(CH, Tl[l] - k * CH)
                if l >= 1:   # input GEMM of this chunk on what the layer below has produced
                    if l == La:
                        src_all = YM[l - 1] if drop > 0.0 else Y[l - 1][1:]
                        src = _stacked(src_all, t0, n, f, B, H)
                    else:
                        src = (YM[l - 1][t0:t0 + n] if drop > 0.0 else Y[l - 1][1 + t0:1 + t0 + n]).reshape(n * B, H)
                    torch.addmm(bias[l], src, Wp[l].t(), out=G[l][t0:t0 + n].view(n * B, 4 * H))
                masked = drop > 0.0 and l < L - 1
                slots.append(_lib.FwdSlot(wt[l].data_ptr(), G[l][t0].data_ptr(), C[l][t0].data_ptr(), Y[l][t0].data_ptr(),
                                          ring[l].data_ptr(), t0 & 1, n, YM[l][t0].data_ptr() if masked else None,
                                          base[l] + t0 * row, drop if masked else 0.0, 0))
            arr = (_lib.FwdSlot * len(slots))(*slots)
            n_launch = max(s_.nsteps for s_ in slots)
            with _lib.timed("lstm_fwd", n_launch, sb * sum(s_.nsteps for s_ in slots)):
                _lib.check(lib.caiman_lstm_wave_fwd(ctypes.cast(arr, ctypes.c_void_p), len(slots), n_launch, B, H, tag,
                                                    int(hard), INTERLEAVED, seed, st))
        saved = [x, Ga, Gb, Ya, Yb, Ca, Cb, *Wp, *Rp]
        if drop > 0.0:
            saved += [YMa, YMb]
        ctx.save_for_backward(*saved)
        ctx.meta = (L, La, f, T1, T2, B, H, hard, drop, seed, x.requires_grad, base)
        ctx.params = params
        y_top = Yb[Lb - 1, 1:]
        all_h_a, all_c_a, all_h_b, all_c_b = Ya[:, 1:T1 + 1], Ca[:, 1:], Yb[:, 1:], Cb[:, 1:]
        ctx.mark_non_differentiable(all_h_a, all_c_a, all_h_b, all_c_b)
        return y_top, all_h_a, all_c_a, all_h_b, all_c_b

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, d_top, *_unused):
        L, La, f, T1, T2, B, H, hard, drop, seed, need_dx, base = ctx.meta
        Lb, T1p = L - La, T2 * f
        from caiman_asr_amd.train_utils import overlap

        overlap.flush_deferred()
        saved = ctx.saved_tensors
        x, Ga, Gb, Ya, Yb, Ca, Cb = saved[:7]
        Wp, Rp = saved[7:7 + L], saved[7 + L:7 + 2 * L]
        YMa, YMb = (saved[7 + 2 * L], saved[8 + 2 * L]) if drop > 0.0 else (None, None)
        dev, dt = Ga.device, Ga.dtype
        tag, lib, st = _lib.dtype_tag(dt), _lib.lib(), _lib.stream()
        row = B * H
        Tl = [T1] * La + [T2] * Lb
        G = [Ga[l] for l in range(La)] + [Gb[m] for m in range(Lb)]
        C = [Ca[l] for l in range(La)] + [Cb[m] for m in range(Lb)]
        dGa, dGb = torch.empty_like(Ga), torch.empty_like(Gb)
        dG = [dGa[l] for l in range(La)] + [dGb[m] for m in range(Lb)]
        if d_top is None:
            d_top = torch.zeros((T2, B, H), dtype=dt, device=dev)
        d_top = d_top.to(dt)
        if d_top.stride(2) != 1:
            d_top = d_top.contiguous()
        # delta[l]: gradient w.r.t. the (masked) output sequence of layer l, filled chunk by chunk
        delta_a = torch.empty((La, T1p, B, H), dtype=dt, device=dev)
        delta_b = torch.empty((max(Lb - 1, 1), T2, B, H), dtype=dt, device=dev)
        delta = [delta_a[l] for l in range(La)] + [delta_b[m] for m in range(Lb - 1)] + [d_top]
        bp = _pad32(B)
        wt = _Scratch.get("ep_bw", L * 4 * H * H, dt, dev).view(L, -1)
        ring = _Scratch.get("ep_br", L * 2 * bp * 4 * H, dt, dev).view(L, -1)
        dC = _Scratch.get("ep_bc", L * B * H, torch.float32, dev).view(L, -1)
        for l in range(L):
            _lib.check(lib.caiman_lstm_prepare(_lib.ptr(Rp[l]), None, _lib.ptr(wt[l]), _lib.ptr(ring[l]), _lib.ptr(dC[l]),
                                                B, H, tag, 1, INTERLEAVED, st))
        nA, nB = (T1 + CH - 1) // CH, (T2 + CH - 1) // CH
        sb = _step_bytes(B, H, Ga.element_size(), True)
        boundary_done = set()   # post chunks whose input gradient has been un-stacked into delta[La-1]
        for tick in reversed(_schedule(nA, nB, La, Lb, f)):
            slots = []
            for l, k in reversed(tick):
                t0, n = k * CH, min(CH, Tl[l] - k * CH)
                thi = t0 + n - 1
                if l == La - 1:      # top pre layer: gradient arrives through StackTime from post layer 0
                    j = t0 // (f * CH)
                    if j not in boundary_done:
                        boundary_done.add(j)
                        p0, pn = j * CH, min(CH, T2 - j * CH)
                        dx2 = torch.matmul(dG[La][p0:p0 + pn].view(pn * B, 4 * H), Wp[La])       # [pn*B, f*H]
                        delta[l][f * p0:f * (p0 + pn)].view(pn, f, B, H).copy_(dx2.view(pn, B, f, H).transpose(1, 2))
                elif l < L - 1:      # dX = dG_{l+1} @ W_{l+1} of the same chunk
                    torch.matmul(dG[l + 1][t0:t0 + n].view(n * B, 4 * H), Wp[l + 1], out=delta[l][t0:t0 + n].view(n * B, H))
                d = delta[l]
                p_slot = drop if l < L - 1 else 0.0
                slots.append(_lib.BwdSlot(wt[l].data_ptr(), G[l][thi].data_ptr(), C[l][thi].data_ptr(), d[thi].data_ptr(),
                                          d.stride(0), d.stride(1), dG[l][thi].data_ptr(), ring[l].data_ptr(),
                                          dC[l].data_ptr(), thi & 1, n, int(thi < Tl[l] - 1), p_slot, base[l] + thi * row))
            arr = (_lib.BwdSlot * len(slots))(*slots)
            n_launch = max(s_.nsteps for s_ in slots)
            with _lib.timed("lstm_bwd", n_launch, sb * sum(s_.nsteps for s_ in slots)):
                _lib.check(lib.caiman_lstm_wave_bwd(ctypes.cast(arr, ctypes.c_void_p), len(slots), n_launch, B, H, tag,
                                                    int(hard), INTERLEAVED, seed, st))

        def layer_input(l):
            if l == 0:
                return x.detach().flatten(0, 1).to(dt)
            if l == La:
                src = YMa[La - 1] if drop > 0.0 else Ya[La - 1, 1:]
                return _stacked(src, 0, T2, f, B, H)
            if l < La:
                return (YMa[l - 1][:T1] if drop > 0.0 else Ya[l - 1, 1:T1 + 1]).reshape(T1 * B, H)
            m = l - La
            return (YMb[m - 1] if drop > 0.0 else Yb[m - 1, 1:]).reshape(T2 * B, H)

        grads = []
        for l in range(L):
            T = Tl[l]
            dg = dG[l].reshape(T * B, 4 * H)
            yprev = (Ya[l, :T1] if l < La else Yb[l - La, :T2]).reshape(T * B, H)
            dB = _unperm_rows(dg.sum(0), H)
            grads += [_unperm_rows(torch.matmul(dg.t(), layer_input(l)), H), _unperm_rows(torch.matmul(dg.t(), yprev), H), dB, dB]
        dX = torch.matmul(dG[0].reshape(T1 * B, 4 * H), Wp[0]).view(T1, B, -1) if need_dx else None
        return (dX, None, None, None, None, None, None, None, None, None, *grads)


def encoder_pipe(x, pre, post, factor, pre_state=None, post_state=None):
    """pre / post: CustomLSTM modules.  -> (y_top [T2,B,H], (all_h_a, all_c_a), (all_h_b, all_c_b))."""
    La, Lb, H = pre.num_layers, post.num_layers, pre.hidden_size
    B = x.shape[1]

    def init(state, L):
        if state is None:
            z = torch.zeros((L, B, H), device=x.device, dtype=x.dtype)
            return z, torch.zeros_like(z)
        return state[0].detach(), state[1].detach()

    h0a, c0a = init(pre_state, La)
    h0b, c0b = init(post_state, Lb)
    params = []
    for mod in (pre, post):
        for layer in mod.layers:
            params += [layer.weight_ih, layer.weight_hh, layer.bias_ih, layer.bias_hh]
    y, aha, aca, ahb, acb = EncoderPipeFunction.apply(x, h0a, c0a, h0b, c0b, pre.hard, float(pre.bl_dropout), pre.training,
                                                      factor, La, *params)
    return y, (aha, aca), (ahb, acb)
