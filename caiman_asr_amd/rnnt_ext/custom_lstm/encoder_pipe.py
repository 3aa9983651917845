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
from caiman_asr_amd.rnnt_ext.custom_lstm.stack import INTERLEAVED, RINGS_ZEROED, _pad32, _Scratch, _unperm_rows

# timesteps per pipeline chunk: 0 = by hidden size (`_chunk`); CAIMAN_ENC_PIPE_CHUNK forces a value
CH = int(__import__("os").environ.get("CAIMAN_ENC_PIPE_CHUNK", "0"))
# post_rnn chunks of CH / factor steps: a post chunk then needs exactly one pre chunk, every post layer works in every
# tick (for CH / factor launches) and the pipeline drains in half the launches; costs smaller input GEMMs
FINE = int(__import__("os").environ.get("CAIMAN_ENC_PIPE_FINE", "1")) != 0
# the chunk GEMMs of the post layers that are active in one tick have the same shape and sit at a constant stride in
# the activation tensors (layer m works one chunk behind layer m-1): issue them as ONE batched GEMM
BMM = int(__import__("os").environ.get("CAIMAN_ENC_PIPE_BMM", "1")) != 0
EARLY_WGRAD = int(__import__("os").environ.get("CAIMAN_EARLY_WGRAD", "1")) != 0
# the chunk GEMMs of a tick (input projections forward, input gradients backward) as ONE launch of the grouped
# projection kernel (csrc/proj_gemm.hip) instead of 2-3 library calls; shapes outside its geometry keep the library path
PROJ = int(__import__("os").environ.get("CAIMAN_PROJ_GEMM", "1")) != 0
PROJ_TILE = int(__import__("os").environ.get("CAIMAN_PROJ_TILE", "0"))
# backward ticks (K = 4H, about one 128 x 128 tile per CU): the variant with a ring of four LDS stages, three in flight
# (tile 10).  In the step 27.19-27.38 -> 26.85-26.97 ms against tile 8 (the workgroup splits K between two wave groups, two
# stages each: 47.5 vs 44.8 us in tools/proj_gemm_bench.py, more in the step where the operands are cold)
PROJ_TILE_BWD = int(__import__("os").environ.get("CAIMAN_PROJ_TILE_BWD", "10"))
IMAGES = int(__import__("os").environ.get("CAIMAN_LSTM_IMAGES", "1")) != 0   # one launch for all operand images of the weights
# the layers' weight gradients dG^T . x / dG^T . h_prev on the transposed-read kernel (csrc/joint_wgrad.hip, fp32 results)
# instead of the library's transposed-A GEMMs; shapes outside its geometry (K = 240 of layer 0) keep the library
WGRAD_TN = int(__import__("os").environ.get("CAIMAN_LSTM_WGRAD_TN", "1")) != 0
WGRAD_BOTH = int(__import__("os").environ.get("CAIMAN_LSTM_WGRAD_BOTH", "1")) != 0   # dR and dW of the post layers in one launch


def _proj_ok(widths, dt):
    """every hidden size (-> N = 4H forward, K = 4H backward) and every K of the stack fits the kernel's tiles."""
    return PROJ and dt in (torch.float16, torch.bfloat16) and all(w % 128 == 0 for w in widths)


_PROJ_DT = None


def _proj_plan(per_tick):
    """per_tick: for every tick a list of caiman_proj_problem_t field tuples.  -> (numpy image of all of them, [(first, count)]
    per tick).  One array for the whole pass: building ctypes structures tick by tick cost 50 us of host time per tick,
    which the pipeline (a launch every ~200 us) does not have to spare."""
    global _PROJ_DT
    import numpy as np

    if _PROJ_DT is None:
        _PROJ_DT = np.dtype([(n_, "<u8") for n_ in ("a", "w", "bias", "c")] +
                            [(n_, "<i4") for n_ in ("M", "N", "K", "a_inner", "a_kseg", "c_inner", "c_nseg", "_pad")] +
                            [(n_, "<i8") for n_ in ("a_so", "a_si", "a_ss", "c_so", "c_si", "c_ss")])
        assert _PROJ_DT.itemsize == ctypes.sizeof(_lib.ProjProblem)
    rows, ranges = [], []
    for probs in per_tick:
        probs.sort(key=lambda r: -r[6])     # longest K first: the long tiles start first, the short ones fill in
        ranges.append((len(rows), len(probs)))
        rows += probs
    if not rows:
        return None, ranges
    plan = np.array(rows, dtype=_PROJ_DT)
    # caiman_proj_gemm() rejects operands it cannot address with 16-byte accesses; every operand here is a buffer this
    # module allocated itself at offsets that are multiples of a row, so a violation is a bug in the plan, not a shape to
    # fall back on: say so now, before the first launch of the pass, instead of CAIMAN_ERR_INVALID in the middle of it
    bad = ((plan["a"] % 16 != 0) | (plan["w"] % 16 != 0) | (plan["c"] % 8 != 0) | (plan["bias"] % 8 != 0) |
           (plan["N"] % 128 != 0) | (plan["K"] % 128 != 0) | (plan["a_kseg"] % 64 != 0) | (plan["c_nseg"] % 16 != 0) |
           ((plan["a_so"] | plan["a_si"] | plan["a_ss"]) % 8 != 0) | ((plan["c_so"] | plan["c_si"] | plan["c_ss"]) % 4 != 0))
    if bad.any():
        raise RuntimeError(f"encoder_pipe: projection problem {plan[bad][0]} is outside caiman_proj_gemm's geometry "
                           "(_proj_ok admitted the widths; operand alignment / segment sizes do not fit)")
    return plan, ranges


def _proj_launch(lib, plan, rng, tag, st, tile):
    first, count = rng
    base = plan.ctypes.data
    for i in range(0, count, 8):
        _lib.check(lib.caiman_proj_gemm(base + (first + i) * _PROJ_DT.itemsize, min(8, count - i), tag, tile, st))


def _skewed(t, first_layer, first_t0, count, n, B, width, chunk, row_offset=0):
    """View [count, n*B, width] of t [layers, steps(+row_offset), B, width]: layer first_layer+i at steps
    [first_t0 - i*chunk, ... + n) -- the operands of `count` consecutive layers of one tick."""
    return torch.as_strided(t, (count, n * B, width), (t.stride(0) - chunk * B * width, width, 1),
                            t[first_layer, row_offset + first_t0].storage_offset())


def eligible(x, hidden, La, Lb, gate_dtype, factor):
    return (x.is_cuda and La >= 1 and Lb >= 1 and La + Lb <= 8 and hidden % 32 == 0 and factor >= 1
            and gate_dtype in (torch.float16, torch.bfloat16))


def _chunk(H):
    """Timesteps per pipeline chunk.  Measured on the final round-3 tree (bench.py, two runs each): H = 1024 (base-85M) 24:
    26.26-26.39 ms, 28: 26.6, 32: 26.5-26.6, 40: 26.7, 48: 26.8, 64: 27.4 (B = 128: 90.1 vs 91.1 at 32); 16 and 20 lose the
    half-length post chunks' alignment and triple the host time.  H = 1536 (large-196M, two launches per tick): 32: 55.7, 24:
    56.3.  Shorter chunks fill and drain the layer pipeline faster; what they cost is one projection + one resident launch
    more per 8 timesteps, which the round-3 projection kernels made cheaper."""
    return CH if CH > 0 else (24 if H <= 1024 else 32)


def _post_chunk(f, ch):
    return ch // f if (FINE and ch % f == 0 and ch // f >= 4) else ch


def _schedule(nA, nB, La, Lb, f, nP=0, Lp=0, ch=32):
    """-> list of ticks, each a list of (layer, chunk).  Layers: pre 0..La-1, post La..La+Lb-1, prediction after."""
    ticks = {}
    for l in range(La):
        for k in range(nA):
            ticks.setdefault(k + l, []).append((l, k))
    per = 1 if _post_chunk(f, ch) != ch else f            # pre chunks stacked into one post chunk
    for m in range(Lb):
        for j in range(nB):
            ready = min(per * j + per - 1, nA - 1) + La   # the last pre chunk it stacks is done at the end of tick (c + La - 1)
            ticks.setdefault(ready + m, []).append((La + m, j))
    for p in range(Lp):                                  # an independent chain that rides in the same launches
        for k in range(nP):
            ticks.setdefault(k + p, []).append((La + Lb + p, k))
    return [sorted(ticks[t]) for t in sorted(ticks)]


def _perm_cast(w, H, dt):
    """rows [gate][unit] -> [unit][gate] and cast, in one copy kernel (stack._perm_rows + .to)."""
    out = torch.empty(w.shape, dtype=dt, device=w.device)
    out.view(H, 4, *w.shape[1:]).copy_(w.view(4, H, *w.shape[1:]).transpose(0, 1))
    return out


def _perm_cast_t(w, H, dt):
    """[4H, K] with rows [gate][unit] -> [K, 4H] with columns [unit][gate], cast: the layout both directions want
    (forward x · Wt is the library's NN kernel, backward dG · Wtᵀ its NT kernel: 41 vs 69 us for the five post layers
    of a tick, tools/lstm_gemm_layout_bench.py)."""
    K = w.shape[1]
    out = torch.empty((K, w.shape[0]), dtype=dt, device=w.device)
    out.view(K, H, 4).copy_(w.view(4, H, K).permute(2, 1, 0))
    return out


def _stacked(src, t0, n, f, B, H):
    """rows of StackTime for post steps [t0, t0+n): src [T1p, B, H] -> [n*B, f*H]."""
    return src[f * t0:f * (t0 + n)].view(n, f, B, H).transpose(1, 2).reshape(n * B, f * H)


class EncoderPipeFunction(torch.autograd.Function):
    """forward(x [T1,B,I], h0a, c0a [La,B,H], h0b, c0b [Lb,B,H], hard, p_drop, training, factor, La, Lb,
               xp [Tp,B,Ip] | None, h0p, c0p [Lp,B,Hp] | None, p_drop_pred, *params)
    -> (y_top [T2,B,H], all_h_a [La,T1,B,H], all_c_a, all_h_b [Lb,T2,B,H], all_c_b, yp_top [Tp,B,Hp], all_h_p, all_c_p)
    params = (W, R, bW, bR) per layer: pre layers, post layers, then the prediction layers (if xp is given: an
    independent LSTM stack -- the prediction network -- whose steps share the encoder's launches; its slots carry
    their own hidden size).  Gradients flow to x, xp and the parameters (states are detached)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, h0a, c0a, h0b, c0b, hard, p_drop, training, factor, La, Lb, xp, h0p, c0p, p_drop_pred, *params):
        L = len(params) // 4
        Le, Lp, f = La + Lb, L - La - Lb, int(factor)
        Ws, Rs, bWs, bRs = params[0::4], params[1::4], params[2::4], params[3::4]
        T1, B, _ = x.shape
        T2 = (T1 + f - 1) // f
        T1p = T2 * f
        Tp = xp.shape[0] if Lp else 0
        dev, lib = x.device, _lib.lib()
        H = Rs[0].shape[1]
        Hp = Rs[Le].shape[1] if Lp else 0
        Hl = [H] * Le + [Hp] * Lp
        Tl = [T1] * La + [T2] * Lb + [Tp] * Lp
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else x.dtype
        tag = _lib.dtype_tag(dt)
        # grouped projection kernel: wants [4H, K] (K contiguous) images of the layers that take a chunk GEMM per tick
        use_proj = _proj_ok(set(Hl) | {f * H}, dt)
        es = torch.empty((), dtype=dt).element_size()
        bp = _pad32(B)
        wt = [_Scratch.get(("ep_fw", l), 4 * Hl[l] * Hl[l], dt, dev) for l in range(L)]
        # every 16-bit image of the fp32 parameters (K-major and N-major input weights, bias sum, forward and backward
        # fragment images of the recurrent weights) in ONE launch (csrc/lstm_images.hip) instead of ~7 small kernels per layer
        fused_img = (IMAGES and dt in (torch.float16, torch.bfloat16) and L <= 16 and
                     all(p_.dtype == torch.float32 and p_.is_contiguous() and p_.is_cuda for p_ in params) and
                     all(W.shape[1] % 4 == 0 and W.data_ptr() % 16 == 0 for W in Ws) and all(R.data_ptr() % 16 == 0 for R in Rs))
        if fused_img:
            Wp = [torch.empty((Ws[l].shape[1], 4 * Hl[l]), dtype=dt, device=dev) for l in range(L)]
            Wn = [(torch.empty((4 * Hl[l], Ws[l].shape[1]), dtype=dt, device=dev) if (use_proj and l != 0 and l != Le) else None)
                  for l in range(L)]
            bias = [torch.empty(4 * Hl[l], dtype=dt, device=dev) for l in range(L)]
            Rp = [torch.empty(4 * Hl[l] * Hl[l], dtype=dt, device=dev) for l in range(L)]   # the BACKWARD images, kept for backward
            imgs = (_lib.LstmImages * L)(*[
                _lib.LstmImages(Ws[l].data_ptr(), Rs[l].data_ptr(), bWs[l].data_ptr(), bRs[l].data_ptr(), Wp[l].data_ptr(),
                                Wn[l].data_ptr() if Wn[l] is not None else None, bias[l].data_ptr(), wt[l].data_ptr(),
                                Rp[l].data_ptr(), Hl[l], Ws[l].shape[1]) for l in range(L)])
            _lib.check(lib.caiman_lstm_weight_images(ctypes.cast(imgs, ctypes.c_void_p), L, tag, _lib.stream()))
        else:
            Rp = [R.to(dt).contiguous() for R in Rs]
            Wp = [_perm_cast_t(Ws[l], Hl[l], dt) for l in range(L)]   # [K, 4H]
            bias = [_perm_cast(bWs[l] + bRs[l], Hl[l], dt) for l in range(L)]
            Wn = [(_perm_cast(Ws[l], Hl[l], dt) if (use_proj and l != 0 and l != Le) else None) for l in range(L)]
        Ga = torch.empty((La, T1, B, 4 * H), dtype=dt, device=dev)
        Gb = torch.empty((Lb, T2, B, 4 * H), dtype=dt, device=dev)
        torch.addmm(bias[0], x.flatten(0, 1).to(dt), Wp[0], out=Ga[0].view(T1 * B, 4 * H))
        # pre outputs carry f-1 zero frames at the end so that the last stacked frame is zero padded (StackTime)
        # (only the padding frames are cleared: the kernels write every other row)
        Ya = torch.empty((La, T1p + 1, B, H), dtype=dt, device=dev)
        if T1p != T1:
            Ya[:, T1 + 1:].zero_()
        Ca = torch.empty((La, T1 + 1, B, H), dtype=dt, device=dev)
        Yb = torch.empty((Lb, T2 + 1, B, H), dtype=dt, device=dev)
        Cb = torch.empty((Lb, T2 + 1, B, H), dtype=dt, device=dev)
        def first_row(buf, state):   # initial state into row 0; None (no carried state): zeros without a tensor of zeros
            buf[:, 0].zero_() if state is None else buf[:, 0].copy_(state)

        first_row(Ya, h0a)
        first_row(Ca, c0a)
        first_row(Yb, h0b)
        first_row(Cb, c0b)
        G = [Ga[l] for l in range(La)] + [Gb[m] for m in range(Lb)]
        Y = [Ya[l] for l in range(La)] + [Yb[m] for m in range(Lb)]
        C = [Ca[l] for l in range(La)] + [Cb[m] for m in range(Lb)]
        Gp = Yp = Cp = None
        if Lp:
            Gp = torch.empty((Lp, Tp, B, 4 * Hp), dtype=dt, device=dev)
            Yp = torch.empty((Lp, Tp + 1, B, Hp), dtype=dt, device=dev)
            Cp = torch.empty((Lp, Tp + 1, B, Hp), dtype=dt, device=dev)
            first_row(Yp, h0p)
            first_row(Cp, c0p)
            torch.addmm(bias[Le], xp.flatten(0, 1).to(dt), Wp[Le], out=Gp[0].view(Tp * B, 4 * Hp))
            G += [Gp[p] for p in range(Lp)]
            Y += [Yp[p] for p in range(Lp)]
            C += [Cp[p] for p in range(Lp)]
        drop_e = float(p_drop) if (training and p_drop > 0.0) else 0.0
        drop_p = float(p_drop_pred) if (training and p_drop_pred > 0.0 and Lp > 1) else 0.0
        pl = [drop_e] * Le + [drop_p] * Lp                    # dropout applied to the OUTPUT of layer l when it feeds a layer
        top = {Le - 1} | ({L - 1} if Lp else set())            # top layers: their output dropout is the caller's
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (drop_e > 0.0 or drop_p > 0.0) else 0
        YMa = torch.empty((La, T1p, B, H), dtype=dt, device=dev) if drop_e > 0.0 else None
        if YMa is not None and T1p != T1:
            YMa[:, T1:].zero_()
        YMb = torch.empty((max(Lb - 1, 1), T2, B, H), dtype=dt, device=dev) if drop_e > 0.0 else None
        YMp = torch.empty((Lp - 1, Tp, B, Hp), dtype=dt, device=dev) if drop_p > 0.0 else None
        YM = [None] * L
        if drop_e > 0.0:
            for l in range(La):
                YM[l] = YMa[l]
            for m in range(Lb - 1):
                YM[La + m] = YMb[m]
        if drop_p > 0.0:
            for p in range(Lp - 1):
                YM[Le + p] = YMp[p]
        base = [0] * L   # dropout counter base of each layer: disjoint ranges
        for l in range(1, L):
            base[l] = base[l - 1] + Tl[l - 1] * B * Hl[l - 1]
        # the h rings of all layers: one buffer, one memset (caiman_lstm_prepare then only tiles the initial state in)
        ring_all = _Scratch.get(("ep_fr_all",), sum(2 * bp * h for h in Hl), dt, dev)
        ring_all.zero_()
        ring, off = [], 0
        for h in Hl:
            ring.append(ring_all[off:off + 2 * bp * h])
            off += 2 * bp * h
        st = _lib.stream()
        zero_state = [h0a is None] * La + [h0b is None] * Lb + [h0p is None] * Lp
        for l in range(L):
            if fused_img and zero_state[l]:
                continue     # nothing to prepare: the weight images exist and a zero initial state is what the cleared ring holds
            _lib.check(lib.caiman_lstm_prepare(None if fused_img else _lib.ptr(Rp[l]), _lib.ptr(Y[l][0]), _lib.ptr(wt[l]),
                                                _lib.ptr(ring[l]), None, B, Hl[l], tag, 0, INTERLEAVED | RINGS_ZEROED, st))
        CH = _chunk(H)
        CHb = _post_chunk(f, CH)
        CHl = [CH] * La + [CHb] * Lb + [CH] * Lp
        nA, nB, nP = (T1 + CH - 1) // CH, (T2 + CHb - 1) // CHb, (Tp + CH - 1) // CH
        sbytes = [_step_bytes(B, h, Ga.element_size(), False) if h else 0 for h in Hl]
        Wt_post = torch.stack([Wp[l] for l in range(La + 1, Le)]) if (BMM and Lb > 2 and not use_proj) else None      # [Lb-1, H, 4H]
        b_post = torch.stack([bias[l] for l in range(La + 1, Le)]).unsqueeze(1) if Wt_post is not None else None
        sched = _schedule(nA, nB, La, Lb, f, nP, Lp, CH)
        if use_proj:
            gp = [g_.data_ptr() for g_ in G]
            # input rows of layer l: the (masked) output of the layer below, one row block [B, H] per timestep
            srcp = [None if (l == 0 or l == Le) else (YM[l - 1].data_ptr() if pl[l - 1] > 0.0 else Y[l - 1][1].data_ptr())
                    for l in range(L)]
            wnp = [w_.data_ptr() if w_ is not None else None for w_ in Wn]
            bp_ = [b_.data_ptr() for b_ in bias]
            per_tick = []
            for tick in sched:
                probs = []
                for l, k in tick:
                    if l == 0 or l == Le:
                        continue
                    t0, n = k * CHl[l], min(CHl[l], Tl[l] - k * CHl[l])
                    hl = Hl[l]
                    c = gp[l] + t0 * B * 4 * hl * es
                    if l == La:      # StackTime: row (t, b) = frames f*t .. f*t + f - 1 of the top pre layer, side by side
                        probs.append((srcp[l] + f * t0 * B * H * es, wnp[l], bp_[l], c, n * B, 4 * hl, f * H,
                                      B, H, n * B, 4 * hl, 0, f * B * H, H, B * H, 0, 4 * hl, 0))
                    else:
                        probs.append((srcp[l] + t0 * B * hl * es, wnp[l], bp_[l], c, n * B, 4 * hl, hl,
                                      n * B, hl, n * B, 4 * hl, 0, 0, hl, 0, 0, 4 * hl, 0))
                per_tick.append(probs)
            plan, ranges = _proj_plan(per_tick)
        for ti, tick in enumerate(sched):
            slots, nbytes = [], 0
            batched = set()
            if use_proj:
                if ranges[ti][1]:
                    _proj_launch(lib, plan, ranges[ti], tag, st, PROJ_TILE)
                batched = {l for l, _ in tick}
            elif Wt_post is not None:   # post layers La+1.. with a full chunk this tick: consecutive layers, chunk index falling by one
                grp = [(l, k) for l, k in tick if La < l < Le and Tl[l] - k * CHb >= CHb]
                if len(grp) >= 2 and all(grp[i + 1][0] == grp[i][0] + 1 and grp[i + 1][1] == grp[i][1] - 1 for i in range(len(grp) - 1)):
                    l0, k0 = grp[0]
                    m0, cnt = l0 - La, len(grp)
                    if pl[l0 - 1] > 0.0:
                        X = _skewed(YMb, m0 - 1, k0 * CHb, cnt, CHb, B, H, CHb)
                    else:
                        X = _skewed(Yb, m0 - 1, k0 * CHb, cnt, CHb, B, H, CHb, row_offset=1)
                    out = _skewed(Gb, m0, k0 * CHb, cnt, CHb, B, 4 * H, CHb)
                    torch.baddbmm(b_post[m0 - 1:m0 - 1 + cnt], X, Wt_post[m0 - 1:m0 - 1 + cnt], out=out)
                    batched = {l for l, _ in grp}
            for l, k in tick:
                t0, n = k * CHl[l], min(CHl[l], Tl[l] - k * CHl[l])
                hl, row = Hl[l], B * Hl[l]
                first = l == 0 or l == Le or l in batched        # first layer of a chain: whole-sequence input GEMM above
                if not first:   # input GEMM of this chunk on what the layer below has produced
                    if l == La:
                        src_all = YM[l - 1] if pl[l - 1] > 0.0 else Y[l - 1][1:]
                        src = _stacked(src_all, t0, n, f, B, H)
                    else:
                        src = (YM[l - 1][t0:t0 + n] if pl[l - 1] > 0.0 else Y[l - 1][1 + t0:1 + t0 + n]).reshape(n * B, hl)
                    torch.addmm(bias[l], src, Wp[l], out=G[l][t0:t0 + n].view(n * B, 4 * hl))
                masked = pl[l] > 0.0 and l not in top
                slots.append(_lib.FwdSlot(wt[l].data_ptr(), G[l][t0].data_ptr(), C[l][t0].data_ptr(), Y[l][t0].data_ptr(),
                                          ring[l].data_ptr(), t0 & 1, n, YM[l][t0].data_ptr() if masked else None,
                                          base[l] + t0 * row, pl[l] if masked else 0.0, hl if hl != H else 0))
                nbytes += n * sbytes[l]
            arr = (_lib.FwdSlot * len(slots))(*slots)
            n_launch = max(s_.nsteps for s_ in slots)
            with _lib.timed("lstm_fwd", n_launch, nbytes) as tm:
                r0 = lib.caiman_lstm_resident_launches() if tm.start is not None else 0
                _lib.check(lib.caiman_lstm_wave_fwd(ctypes.cast(arr, ctypes.c_void_p), len(slots), n_launch, B, H, tag,
                                                    int(hard), INTERLEAVED, seed, st))
                if tm.start is not None and lib.caiman_lstm_resident_launches() != r0:
                    # one resident launch: every slot reads its recurrent matrix once, not once per timestep
                    tm.units = 1
                    tm.nbytes = nbytes - sum((s_.nsteps - 1) * 4 * (s_.hidden or H) ** 2 * Ga.element_size() for s_ in slots)
        saved = [x, Ga, Gb, Ya, Yb, Ca, Cb, *Wp, *Rp]
        flags = (drop_e > 0.0, bool(Lp), drop_p > 0.0, fused_img)
        if drop_e > 0.0:
            saved += [YMa, YMb]
        if Lp:
            saved += [xp, Gp, Yp, Cp]
        if drop_p > 0.0:
            saved += [YMp]
        ctx.save_for_backward(*saved)
        ctx.params = params
        ctx.meta = (L, La, Lb, f, T1, T2, Tp, B, H, Hp, hard, pl, seed, x.requires_grad, Lp and xp.requires_grad, base, flags)
        y_top = Yb[Lb - 1, 1:]
        all_h_a, all_c_a, all_h_b, all_c_b = Ya[:, 1:T1 + 1], Ca[:, 1:], Yb[:, 1:], Cb[:, 1:]
        if Lp:
            yp_top, all_h_p, all_c_p = Yp[Lp - 1, 1:], Yp[:, 1:], Cp[:, 1:]
        else:
            yp_top = all_h_p = all_c_p = y_top.new_zeros(0)
        ctx.mark_non_differentiable(all_h_a, all_c_a, all_h_b, all_c_b, all_h_p, all_c_p)
        return y_top, all_h_a, all_c_a, all_h_b, all_c_b, yp_top, all_h_p, all_c_p

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, d_top, _a, _b, _c, _d, d_top_p, *_unused):
        L, La, Lb, f, T1, T2, Tp, B, H, Hp, hard, pl, seed, need_dx, need_dxp, base, flags = ctx.meta
        Le, Lp, T1p = La + Lb, L - La - Lb, T2 * f
        from caiman_asr_amd.train_utils import overlap

        overlap.fence_collectives()   # resident launches and a collective's kernel must not be placed by halves together
        saved = list(ctx.saved_tensors)
        x, Ga, Gb, Ya, Yb, Ca, Cb = saved[:7]
        Wp, Rp = saved[7:7 + L], saved[7 + L:7 + 2 * L]
        rest = saved[7 + 2 * L:]
        YMa = YMb = xp = Gp = Yp = Cp = YMp = None
        if flags[0]:
            YMa, YMb, rest = rest[0], rest[1], rest[2:]
        if flags[1]:
            xp, Gp, Yp, Cp, rest = rest[0], rest[1], rest[2], rest[3], rest[4:]
        if flags[2]:
            YMp = rest[0]
        dev, dt = Ga.device, Ga.dtype
        tag, lib, st = _lib.dtype_tag(dt), _lib.lib(), _lib.stream()
        Hl = [H] * Le + [Hp] * Lp
        Tl = [T1] * La + [T2] * Lb + [Tp] * Lp
        top = {Le - 1} | ({L - 1} if Lp else set())
        G = [Ga[l] for l in range(La)] + [Gb[m] for m in range(Lb)] + ([Gp[p] for p in range(Lp)] if Lp else [])
        C = [Ca[l] for l in range(La)] + [Cb[m] for m in range(Lb)] + ([Cp[p] for p in range(Lp)] if Lp else [])
        dGa, dGb = torch.empty_like(Ga), torch.empty_like(Gb)
        dGp = torch.empty_like(Gp) if Lp else None
        dG = [dGa[l] for l in range(La)] + [dGb[m] for m in range(Lb)] + ([dGp[p] for p in range(Lp)] if Lp else [])

        def as_delta(d, T, h):
            if d is None:
                d = torch.zeros((T, B, h), dtype=dt, device=dev)
            d = d.to(dt)
            return d if d.stride(2) == 1 else d.contiguous()

        # delta[l]: gradient w.r.t. the (masked) output sequence of layer l, filled chunk by chunk
        delta_a = torch.empty((La, T1p, B, H), dtype=dt, device=dev)
        delta_b = torch.empty((max(Lb - 1, 1), T2, B, H), dtype=dt, device=dev)
        delta = [delta_a[l] for l in range(La)] + [delta_b[m] for m in range(Lb - 1)] + [as_delta(d_top, T2, H)]
        if Lp:
            delta_p = torch.empty((max(Lp - 1, 1), Tp, B, Hp), dtype=dt, device=dev)
            delta += [delta_p[p] for p in range(Lp - 1)] + [as_delta(d_top_p, Tp, Hp)]
        bp = _pad32(B)
        # flags[3]: the forward pass saved the backward fragment images themselves (caiman_lstm_weight_images)
        wt = list(Rp) if flags[3] else [_Scratch.get(("ep_bw", l), 4 * Hl[l] * Hl[l], dt, dev) for l in range(L)]
        # dG rings (16-bit) and dC carries (fp32) of all layers in one byte buffer: one memset instead of two per layer
        es_ = Ga.element_size()
        ring_b = [2 * bp * 4 * h * es_ for h in Hl]
        dc_b = [B * h * 4 for h in Hl]
        zero_all = _Scratch.get(("ep_bz_all",), sum(ring_b) + sum(dc_b) + 16 * L, torch.uint8, dev)
        zero_all.zero_()
        ring, dC, off = [], [], 0
        for l in range(L):
            ring.append(zero_all[off:off + ring_b[l]].view(dt))
            off += (ring_b[l] + 15) // 16 * 16
        for l in range(L):
            dC.append(zero_all[off:off + dc_b[l]].view(torch.float32))
            off += (dc_b[l] + 15) // 16 * 16
        for l in range(L):
            if flags[3]:
                continue     # images saved by the forward pass, rings and carries cleared above: nothing left to prepare
            _lib.check(lib.caiman_lstm_prepare(_lib.ptr(Rp[l]), None, _lib.ptr(wt[l]), _lib.ptr(ring[l]),
                                                _lib.ptr(dC[l]), B, Hl[l], tag, 1, INTERLEAVED | RINGS_ZEROED, st))
        CH = _chunk(H)
        CHb = _post_chunk(f, CH)
        CHl = [CH] * La + [CHb] * Lb + [CH] * Lp
        nA, nB, nP = (T1 + CH - 1) // CH, (T2 + CHb - 1) // CHb, (Tp + CH - 1) // CH
        sbytes = [_step_bytes(B, h, Ga.element_size(), True) if h else 0 for h in Hl]
        # bias gradients: the backward kernels add each launch's dG row sums here (BwdSlot.dbias), no pass over dG afterwards
        # (only where the weight-resident kernels run: they keep the sums in registers; the per-timestep path would need an
        # extra reduction launch per call, slower than one batched sum at the end)
        fused_db = all(lib.caiman_lstm_resident_would_run(B, h, min(8, L)) for h in set(Hl))
        dbias = torch.zeros((L, 4 * max(Hl)), dtype=torch.float32, device=dev) if fused_db else None
        boundary_done = set()   # post chunks whose input gradient has been un-stacked into delta[La-1]
        use_proj = _proj_ok(set(Hl) | {f * H}, dt)
        W_post = torch.stack([Wp[l] for l in range(La + 1, Le)]).transpose(1, 2) if (BMM and Lb > 2 and not use_proj) else None   # views [Lb-1, 4H, H]
        sched = list(reversed(_schedule(nA, nB, La, Lb, f, nP, Lp, CH)))
        if use_proj:
            # input gradients of a tick's chunks: delta_l = dG_{l+1} @ W_{l+1} (W stored [K_in, 4H] = the kernel's [N][K]
            # operand), the top pre layer's through StackTime (the columns of a row scatter to f frames)
            es = Ga.element_size()
            dgp = [g_.data_ptr() for g_ in dG]
            dlp = [d_.data_ptr() for d_ in delta]
            wpp = [w_.data_ptr() for w_ in Wp]
            per_tick = []
            for tick in sched:
                probs = []
                for l, k in reversed(tick):
                    if l in top:
                        continue
                    t0, n = k * CHl[l], min(CHl[l], Tl[l] - k * CHl[l])
                    hl = Hl[l]
                    if l == La - 1:
                        j = t0 // (f * CHb)
                        if j in boundary_done:
                            continue
                        boundary_done.add(j)
                        p0, pn = j * CHb, min(CHb, T2 - j * CHb)
                        probs.append((dgp[La] + p0 * B * 4 * H * es, wpp[La], 0, dlp[l] + f * p0 * B * H * es, pn * B, f * H, 4 * H,
                                      pn * B, 4 * H, B, H, 0, 0, 4 * H, 0, f * B * H, H, B * H))
                    else:
                        probs.append((dgp[l + 1] + t0 * B * 4 * hl * es, wpp[l + 1], 0, dlp[l] + t0 * B * hl * es, n * B, hl, 4 * hl,
                                      n * B, 4 * hl, n * B, hl, 0, 0, 4 * hl, 0, 0, hl, 0))
                per_tick.append(probs)
            plan, ranges = _proj_plan(per_tick)
        for ti, tick in enumerate(sched):
            slots, nbytes = [], 0
            batched = set()
            if use_proj:
                if ranges[ti][1]:
                    _proj_launch(lib, plan, ranges[ti], tag, st, PROJ_TILE_BWD)
                batched = {l for l, _ in tick}
            elif W_post is not None:   # delta of post layers La..Le-2 with a full chunk: dG of the layer above times its W_ih
                grp = [(l, k) for l, k in tick if La <= l < Le - 1 and Tl[l] - k * CHb >= CHb]
                if len(grp) >= 2 and all(grp[i + 1][0] == grp[i][0] + 1 and grp[i + 1][1] == grp[i][1] - 1 for i in range(len(grp) - 1)):
                    l0, k0 = grp[0]
                    m0, cnt = l0 - La, len(grp)
                    X = _skewed(dGb, m0 + 1, k0 * CHb, cnt, CHb, B, 4 * H, CHb)
                    out = _skewed(delta_b, m0, k0 * CHb, cnt, CHb, B, H, CHb)
                    torch.bmm(X, W_post[m0:m0 + cnt], out=out)
                    batched = {l for l, _ in grp}
            for l, k in reversed(tick):
                t0, n = k * CHl[l], min(CHl[l], Tl[l] - k * CHl[l])
                thi, hl, row = t0 + n - 1, Hl[l], B * Hl[l]
                if l in batched:
                    pass             # the grouped projection kernel has written delta[l] of this chunk
                elif l == La - 1:    # top pre layer: gradient arrives through StackTime from post layer 0
                    j = t0 // (f * CHb)
                    if j not in boundary_done:
                        boundary_done.add(j)
                        p0, pn = j * CHb, min(CHb, T2 - j * CHb)
                        dx2 = torch.matmul(dG[La][p0:p0 + pn].view(pn * B, 4 * H), Wp[La].t())   # [pn*B, f*H]
                        delta[l][f * p0:f * (p0 + pn)].view(pn, f, B, H).copy_(dx2.view(pn, B, f, H).transpose(1, 2))
                elif l not in top and l not in batched:   # dX = dG_{l+1} @ W_{l+1} of the same chunk
                    torch.matmul(dG[l + 1][t0:t0 + n].view(n * B, 4 * hl), Wp[l + 1].t(), out=delta[l][t0:t0 + n].view(n * B, hl))
                d = delta[l]
                p_slot = pl[l] if l not in top else 0.0
                slots.append(_lib.BwdSlot(wt[l].data_ptr(), G[l][thi].data_ptr(), C[l][thi].data_ptr(), d[thi].data_ptr(),
                                          d.stride(0), d.stride(1), dG[l][thi].data_ptr(), ring[l].data_ptr(),
                                          dC[l].data_ptr(), thi & 1, n, int(thi < Tl[l] - 1), p_slot, base[l] + thi * row,
                                          hl if hl != H else 0, 0, dbias[l].data_ptr() if fused_db else None))
                nbytes += n * sbytes[l]
            arr = (_lib.BwdSlot * len(slots))(*slots)
            n_launch = max(s_.nsteps for s_ in slots)
            with _lib.timed("lstm_bwd", n_launch, nbytes) as tm:
                r0 = lib.caiman_lstm_resident_launches() if tm.start is not None else 0
                _lib.check(lib.caiman_lstm_wave_bwd(ctypes.cast(arr, ctypes.c_void_p), len(slots), n_launch, B, H, tag,
                                                    int(hard), INTERLEAVED, seed, st))
                if tm.start is not None and lib.caiman_lstm_resident_launches() != r0:
                    # one resident launch: every slot reads its recurrent matrix once, not once per timestep
                    tm.units = 1
                    tm.nbytes = nbytes - sum((s_.nsteps - 1) * 4 * (s_.hidden or H) ** 2 * Ga.element_size() for s_ in slots)

        def layer_input(l):
            if l == 0:
                return x.detach().flatten(0, 1).to(dt)
            if l == Le:
                return xp.detach().flatten(0, 1).to(dt)
            if l == La:
                src = YMa[La - 1] if pl[La - 1] > 0.0 else Ya[La - 1, 1:]
                return _stacked(src, 0, T2, f, B, H)
            if l < La:
                return (YMa[l - 1][:T1] if pl[l - 1] > 0.0 else Ya[l - 1, 1:T1 + 1]).reshape(T1 * B, H)
            if l < Le:
                m = l - La
                return (YMb[m - 1] if pl[l - 1] > 0.0 else Yb[m - 1, 1:]).reshape(T2 * B, H)
            p = l - Le
            return (YMp[p - 1] if pl[l - 1] > 0.0 else Yp[p - 1, 1:]).reshape(Tp * B, Hp)

        # Data-parallel runs: every layer's backward ends in the last few ticks, so nothing can be reduced earlier --
        # but the ~3 ms of weight-gradient GEMMs that follow can hide the collectives.  With a gradient reducer listening
        # (train_utils/overlap.py callbacks) each layer's gradients go into `.grad` as soon as they exist, top layers
        # first (the reducer cuts its buckets from the tail of the arena), and autograd gets None for them.
        # Weight gradients leave this function by being ADDED into `param.grad` (a view of the optimiser's fp32 arena),
        # un-permuting the gate rows and widening to fp32 in that one kernel, instead of being returned: autograd would
        # add them anyway, after a separate un-permute copy and a cast.  Listeners (the data-parallel reducer,
        # train_utils/overlap.py) are told per parameter, top layers first: every layer's backward ends in the last
        # few ticks, so nothing can be reduced earlier than here, but the collectives then run under the remaining
        # weight-gradient GEMMs.
        direct = EARLY_WGRAD

        def deliver(param, g, hl):
            """g: gradient with rows in the pipeline's [unit][gate] order -> param.grad (rows [gate][unit])."""
            if not param.requires_grad:
                return
            if param.grad is None:
                param.grad = torch.zeros_like(param)
            param.grad.view(4, hl, *param.shape[1:]).add_(g.view(hl, 4, *g.shape[1:]).transpose(0, 1))
            overlap.notify_grad_ready(param)

        # post layers have identical shapes: their recurrent-weight gradients (all Lb), their input-weight gradients
        # (layers 1..Lb-1) and all bias gradients are three batched calls instead of 3 * Lb
        # the delivery kernel adds the weight-gradient kernel's partial products (slabs) up on its way into `.grad`: no
        # separate reduction pass.  Only when every layer's gradients leave through it (`direct`, fp32 parameters).
        raw = (direct and IMAGES and all(p_.requires_grad and p_.dtype == torch.float32 and p_.is_contiguous() for p_ in ctx.params))

        def tn(dg3, x3):
            """dg3 [P, rows, 4H]^T . x3 [P, rows, K] -> [P, 4H, K] (or the slabs [P, slices, 4H, K] with `raw`): the
            hand-written kernel where it applies, else the library"""
            out = overlap.wgrad_tn(dg3, x3, only_if_faster=True, raw=raw) if WGRAD_TN else None
            return out if out is not None else torch.bmm(dg3.transpose(1, 2), x3, out_dtype=torch.float32)   # fp32 products, like the kernel's

        def rows3(t, first, count, skip, T):
            """layers [first, first + count) of t [layers, T (+ 1), B, H], steps [skip, skip + T) -> [count, T * B, H] without
            a copy (rows contiguous, layers a constant stride apart)"""
            v = t[first:first + count, skip:skip + T]
            return torch.as_strided(v, (count, T * B, v.shape[-1]), (t.stride(0), v.shape[-1], 1), v.storage_offset())

        post_R = post_W = post_b = None
        if BMM and Lb > 1:
            dgb = dGb.view(Lb, T2 * B, 4 * H)
            xin = rows3(YMb, 0, Lb - 1, 0, T2) if pl[La] > 0.0 else rows3(Yb, 0, Lb - 1, 1, T2)
            both = None
            if WGRAD_TN and WGRAD_BOTH:
                # dR of all Lb layers and dW of Lb - 1: one shape, two activation buffers.  As ONE launch when the cost model
                # prefers that to two (at B = 32: 704 tiles = 2.75 rounds of the chip in one slice, where five layers alone
                # would leave their last round half empty and go to the library)
                est = overlap.wgrad_tn_estimate_us
                rows, e2 = T2 * B, None
                e_both, e_r, e_w = est(rows, 4 * H, H, 2 * Lb - 1, dt), est(rows, 4 * H, H, Lb, dt), est(rows, 4 * H, H, Lb - 1, dt)
                lib_us = lambda n: 2.0 * n * rows * 4 * H * H / overlap.LIBRARY_TN_FLOPS * 1e6
                if e_both > 0:
                    e2 = min(e_r, lib_us(Lb)) if e_r > 0 else lib_us(Lb)
                    e2 += min(e_w, lib_us(Lb - 1)) if e_w > 0 else lib_us(Lb - 1)
                    if e_both < e2:
                        both = overlap.wgrad_tn(dgb, rows3(Yb, 0, Lb, 0, T2), second=(dgb[1:], xin), raw=raw)
            if both is not None:
                post_R, post_W = both[:Lb], both[Lb:]
            else:
                post_R = tn(dgb, rows3(Yb, 0, Lb, 0, T2))
                post_W = tn(dgb[1:], xin)
            post_b = None if fused_db else dgb.sum(1, dtype=torch.float32)
        # pre layers: dR of all La layers and dW of layers 1 .. La - 1 have one shape (layer 0's dW has K = in_feats): one
        # launch of the weight-gradient kernel where its cost model prefers that to one product at a time
        pre_R = pre_W = None
        if WGRAD_TN and WGRAD_BOTH and La >= 2 and all(Hl[l] == H for l in range(La)):
            est = overlap.wgrad_tn_estimate_us
            rows = T1 * B
            e_all, e_one = est(rows, 4 * H, H, 2 * La - 1, dt), est(rows, 4 * H, H, 1, dt)
            lib_one = 2.0 * rows * 4 * H * H / overlap.LIBRARY_TN_FLOPS * 1e6
            if e_all > 0 and e_all < (2 * La - 1) * (min(e_one, lib_one) if e_one > 0 else lib_one):
                dga = dGa.view(La, T1 * B, 4 * H)
                xin_a = rows3(YMa, 0, La - 1, 0, T1) if pl[0] > 0.0 else rows3(Ya, 0, La - 1, 1, T1)
                got = overlap.wgrad_tn(dga, rows3(Ya, 0, La, 0, T1), second=(dga[1:], xin_a), raw=raw)
                if got is not None:
                    pre_R, pre_W = got[:La], got[La:]
        per_layer = [None] * L
        for l in (reversed(range(L)) if direct else range(L)):
            T, hl = Tl[l], Hl[l]
            dg = dG[l].reshape(T * B, 4 * hl)
            m = l - La
            if post_R is not None and 0 <= m < Lb:
                dB = dbias[l, :4 * hl] if fused_db else post_b[m]
                gW = post_W[m - 1] if m >= 1 else tn(dg.unsqueeze(0), layer_input(l).unsqueeze(0))[0]
                g4 = [gW, post_R[m], dB, dB]
            elif pre_R is not None and l < La:
                dB = dbias[l, :4 * hl] if fused_db else dg.sum(0, dtype=torch.float32)
                gW = pre_W[l - 1] if l >= 1 else tn(dg.unsqueeze(0), layer_input(l).unsqueeze(0))[0]
                g4 = [gW, pre_R[l], dB, dB]
            else:
                yprev = (Ya[l, :T1] if l < La else Yb[l - La, :T2] if l < Le else Yp[l - Le, :Tp]).reshape(T * B, hl)
                dB = dbias[l, :4 * hl] if fused_db else dg.sum(0, dtype=torch.float32)
                g4 = [tn(dg.unsqueeze(0), layer_input(l).unsqueeze(0))[0], tn(dg.unsqueeze(0), yprev.unsqueeze(0))[0], dB, dB]
            if direct:
                ps = ctx.params[4 * l:4 * l + 4]
                if (IMAGES and all(p_.requires_grad and p_.dtype == torch.float32 and p_.is_contiguous() for p_ in ps)
                        and all(g_.is_contiguous() and g_.dtype in (dt, torch.float32) for g_ in g4)):
                    # the four parameters of the layer in one launch (un-permute + widen + accumulate)
                    for p_ in ps:
                        if p_.grad is None:
                            p_.grad = torch.zeros_like(p_)
                    # a 3-D weight gradient is a stack of slabs (wgrad_tn(raw=True)): the kernel sums them
                    items = (_lib.GradItem * 4)(*[
                        _lib.GradItem(g_.data_ptr(), p_.grad.data_ptr(), hl, p_.shape[1] if p_.dim() == 2 else 1,
                                      int(g_.dtype == torch.float32), g_.shape[0] if g_.dim() == 3 else 1)
                        for p_, g_ in zip(ps, g4)])
                    _lib.check(lib.caiman_lstm_grad_deliver(ctypes.cast(items, ctypes.c_void_p), 4, tag, st))
                    for p_ in ps:
                        overlap.notify_grad_ready(p_)
                else:
                    for p_, g_ in zip(ps, g4):
                        deliver(p_, g_.sum(0) if (g_.dim() == 3 and p_.dim() == 2) else g_, hl)
                g4 = [None] * 4
            else:
                g4 = [_unperm_rows(g_, hl) for g_ in g4]
            per_layer[l] = g4
        grads = [g for g4 in per_layer for g in g4]
        dX = torch.matmul(dG[0].reshape(T1 * B, 4 * H), Wp[0].t()).view(T1, B, -1) if need_dx else None
        dXp = torch.matmul(dG[Le].reshape(Tp * B, 4 * Hp), Wp[Le].t()).view(Tp, B, -1) if (Lp and need_dxp) else None
        return (dX, None, None, None, None, None, None, None, None, None, None, dXp, None, None, None, *grads)


def _states(state, L, B, H, like):
    if state is None:
        return None, None          # the function clears the first row of its own buffers
    return state[0].detach(), state[1].detach()


def encoder_pipe(x, pre, post, factor, pre_state=None, post_state=None, pred=None, xp=None, pred_state=None):
    """pre / post (/ pred): CustomLSTM modules; xp [Tp, B, Ip]: the prediction network's input sequence.
    -> (y_top [T2,B,H], (all_h_a, all_c_a), (all_h_b, all_c_b), yp_top | None, (all_h_p, all_c_p) | None)."""
    La, Lb, H, B = pre.num_layers, post.num_layers, pre.hidden_size, x.shape[1]
    h0a, c0a = _states(pre_state, La, B, H, x)
    h0b, c0b = _states(post_state, Lb, B, H, x)
    mods = [pre, post]
    h0p = c0p = None
    if pred is not None:
        h0p, c0p = _states(pred_state, pred.num_layers, B, pred.hidden_size, x)
        mods.append(pred)
    params = []
    for mod in mods:
        for layer in mod.layers:
            params += [layer.weight_ih, layer.weight_hh, layer.bias_ih, layer.bias_hh]
    y, aha, aca, ahb, acb, yp, ahp, acp = EncoderPipeFunction.apply(
        x, h0a, c0a, h0b, c0b, pre.hard, float(pre.bl_dropout), pre.training, factor, La, Lb,
        xp if pred is not None else None, h0p, c0p, float(pred.bl_dropout) if pred is not None else 0.0, *params)
    if pred is None:
        return y, (aha, aca), (ahb, acb), None, None
    return y, (aha, aca), (ahb, acb), yp, (ahp, acp)
