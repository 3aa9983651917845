"""GPU parity of the hand-written joint projection GEMM with the row log-sum-exp in its epilogue (csrc/joint_gemm.hip,
caiman_joint_fc_forward) against plain fp32 torch evaluations of the same products
(training/caiman_asr_train/rnnt/model.py:409-439 `joint_fc`; training/lib/csrc/logsumexp.cu:65-105 for the normaliser).

Tolerances: operands are 16-bit, accumulation fp32 (K <= 8704 terms), the result is rounded once to the storage type:
|c - ref| <= 1 ulp of the storage type at |ref| (+ the fp32 summation-order noise, 1e-3 absolute at these magnitudes).
The log-sum-exp is taken over the STORED row, so it is compared with torch.logsumexp of the kernel's own output in fp32:
what is left is the hardware exp (1e-6 relative) and the order of 8704 additions."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(a, w, bias, want_lse):
    from caiman_asr_amd.train_utils.overlap import _joint_gemm

    out = _joint_gemm(a, w, bias, want_lse)
    assert out is not None, "shape rejected by caiman_joint_fc_supported"
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,bias,lse", [
    (1000, 512, 256, True, True),        # ragged last M tile, two N tiles, the shortest K (four K tiles: first + last pair only)
    (257, 8704, 768, True, True),        # the joint projection's own N and K; one full and one 1-row tile
    (256, 256, 256, False, True),        # a single tile
    (193, 256, 384, True, True),         # one tile whose second A half (rows 64 ..) ends inside the tile, its fourth not at all
    (4101, 768, 8704, False, False),     # the input gradient dY . W: N = 768, K = 8704 (136 K tiles)
    (3000, 17408, 1024, True, True),     # large-196M: V = 17 408, joint_n_hid = 1024
    (70000, 1024, 512, True, True),      # 1096 tiles on 256 persistent workgroups: 4-5 tiles each, ragged last M tile
])
def test_joint_fc_gemm_matches_fp32_product_and_row_lse(dtype, M, N, K, bias, lse):
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = torch.randn(M, K, device=DEV, generator=g).to(dtype)
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5 * 3.0).to(dtype)
    b = (torch.randn(N, device=DEV, generator=g)).to(dtype) if bias else None
    c, l = _run(a, w, b, lse)
    ref = a.float() @ w.float().t()
    if bias:
        ref = ref + b.float()
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10          # one ulp relative (8 / 11 significant bits)
    err = (c.float() - ref).abs()
    assert (err <= ulp * ref.abs() + 2e-3).all(), float((err - ulp * ref.abs()).max())
    assert err.mean() <= 0.3 * ulp * ref.abs().mean() + 1e-4
    if lse:
        want = torch.logsumexp(c.float(), dim=1)
        assert l.dtype == torch.float32 and l.shape == (M,)
        assert torch.allclose(l, want, atol=2e-4, rtol=1e-6), float((l - want).abs().max())
    else:
        assert l is None


def test_joint_fc_gemm_is_deterministic_over_repeated_launches():
    """Race screen for the four-stage LDS-DMA ring (a read placed before the wait + barrier that retire the stage's DMAs
    passes reference checks whenever the DMA happens to land first): the joint projection at training size, 12 launches,
    bit-identical logits and normalisers; the chip is kept busy by a second stream to vary the timing."""
    M, N, K = 40000, 8704, 768
    g = torch.Generator(device=DEV).manual_seed(5)
    a = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=DEV, generator=g).to(torch.bfloat16)
    c0, l0 = _run(a, w, b, True)
    ref = (a[:512].float() @ w.float().t() + b.float())
    assert ((c0[:512].float() - ref).abs() <= 2.0 ** -7 * ref.abs() + 2e-3).all()
    side = torch.cuda.Stream()
    x = torch.randn(4096, 4096, device=DEV)
    for i in range(12):
        if i % 2:
            with torch.cuda.stream(side):
                for _ in range(4):
                    x @ x
        c, l = _run(a, w, b, True)
        assert torch.equal(c, c0) and torch.equal(l, l0), i


def test_nan_rows_propagate_to_their_normalisers_only():
    M, N, K = 300, 512, 256
    a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    a[7, 3] = float("nan")
    w = torch.randn(N, K, device=DEV).to(torch.bfloat16)
    c, l = _run(a, w, None, True)
    assert torch.isnan(l[7]) and torch.isnan(c[7]).all()
    assert torch.isfinite(l[torch.arange(M, device=DEV) != 7]).all()


def test_model_step_with_the_handwritten_joint_projection_equals_the_library_path(monkeypatch):
    """One bf16 training step of the golden mini model... its joint sizes (V = 29, Hj = 32) are outside the kernel's
    geometry, so a model with N % 256 == 0 and K = 256 is built here: loss and gradients with CAIMAN_JOINT_GEMM on
    and off agree to the storage resolution, and the loss really took the normalisers from the projection."""
    from caiman_asr_amd.rnnt.loss import ApexTransducerLoss, get_packing_meta_data
    from caiman_asr_amd.rnnt.model import RNNT
    from caiman_asr_amd.rnnt_ext.transducer import loss as tl
    from caiman_asr_amd.train_utils import overlap

    cfg = dict(in_feats=48, enc_n_hid=128, enc_pre_rnn_layers=1, enc_post_rnn_layers=1, enc_stack_time_factor=2,
               enc_dropout=0.0, enc_batch_norm=False, enc_freeze=False, pred_n_hid=128, pred_rnn_layers=1, pred_dropout=0.0,
               pred_batch_norm=False, joint_n_hid=256, joint_dropout=0.0, joint_net_lr_factor=1.0, joint_apex_transducer="pack",
               joint_apex_relu_dropout=True, forget_gate_bias=1.0, custom_lstm=True, quantize=False, enc_rw_dropout=0.0,
               pred_rw_dropout=0.0)
    V, B, T = 512, 6, 24
    torch.manual_seed(9)
    m = RNNT(n_classes=V, **cfg).to(DEV).train()
    x = torch.randn(T, B, 48, device=DEV)
    xl = torch.tensor([24, 20, 24, 11, 24, 17])
    y = torch.randint(0, V - 1, (B, 5), device=DEV)
    yl = torch.tensor([5, 3, 4, 5, 1, 2])
    meta = get_packing_meta_data(xl, yl, 2, device=DEV)
    loss_fn = ApexTransducerLoss(blank_idx=V - 1, eos_idx=None, star_idx=None, packed_input=True)
    taken, bias_sums = [], []
    real_take, real_bias = tl.take_row_lse, tl.take_bias_gradient
    monkeypatch.setattr(tl, "take_row_lse", lambda t: (taken.append(real_take(t)), taken[-1])[1])
    monkeypatch.setattr(tl, "take_bias_gradient", lambda t: (bias_sums.append(real_bias(t)), bias_sums[-1])[1])
    out = []
    for mode in ("1", "0"):
        monkeypatch.setattr(overlap, "JOINT_GEMM", mode)
        m.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits, out_lens, _ = m(x, xl.to(DEV), y, yl.to(DEV), batch_offset=meta["batch_offset"], packed_batch=meta["packed_batch"])
            loss = loss_fn(logits, out_lens, y, yl.to(DEV), meta["batch_offset"], meta["max_f_len"])
        loss.backward()
        out.append((float(loss), {n: p.grad.float().clone() for n, p in m.named_parameters()}))
    assert taken[0] is not None and taken[1] is None      # hand-written path: normalisers from the GEMM; library path: none offered
    # the loss backward's fused column sums reach the projection's bias gradient in both modes (the hand-over checks storage,
    # shape, dtype and the version counter of the gradient tensor: a miss would silently cost a 5 GB pass per step)
    assert len(bias_sums) == 2 and all(b is not None for b in bias_sums)
    (l1, g1), (l0, g0) = out
    assert abs(l1 - l0) <= 2e-3 * abs(l0), (l1, l0)
    for n in g0:
        scale = float(g0[n].abs().max()) + 1e-12
        assert float((g1[n] - g0[n]).abs().max()) <= 2e-2 * scale, n


# ---- weight gradient: dW = dY^T . h on the transposed-read kernel (csrc/joint_wgrad.hip) ----------------------------------
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [
    (512, 256, 256),          # one tile, the shortest slice (4 stages of 32 rows per slice at most)
    (5000, 512, 768),         # rows past the slices (5000 is not a multiple of 128): they ride in the last slice, zero-padded by the descriptor
    (20011, 8704, 768),       # the joint projection's own N and K: 102 tiles x 5 slices
    (9000, 2048, 1024),       # K = 1024 (large-196M joint_n_hid)
])
def test_joint_fc_wgrad_matches_fp32_product(dtype, M, N, K):
    """fp32 accumulation of 16-bit operands; the result is fp32, so what differs from the fp32 reference product is the
    order of M additions: |err| <= 1e-5 * sqrt(M) * the operands' scale (measured 3e-6 relative to the result's range)."""
    from caiman_asr_amd.train_utils.overlap import _joint_wgrad

    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    dy = torch.randn(M, N, device=DEV, generator=g).to(dtype)
    h = torch.randn(M, K, device=DEV, generator=g).to(dtype)
    h[:, 3] = 0.0                                   # asymmetric content: a swapped n / k would not pass
    dy[:, 5] *= 4.0
    dw = _joint_wgrad(dy, h)
    assert dw is not None, "shape rejected by caiman_joint_fc_wgrad_plan"
    torch.cuda.synchronize()
    assert dw.dtype == torch.float32 and dw.shape == (N, K)
    ref = torch.zeros(N, K, device=DEV, dtype=torch.float64)
    for m0 in range(0, M, 4096):
        ref += dy[m0:m0 + 4096].double().t() @ h[m0:m0 + 4096].double()
    err = (dw.double() - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item(), (err, ref.abs().max().item())
    assert (dw[:, 3] == 0).all()


def test_joint_fc_wgrad_is_deterministic_over_repeated_launches():
    """Race screen for the ring (see the forward's): 8 launches at training width, bit-identical slabs."""
    from caiman_asr_amd.train_utils.overlap import _joint_wgrad

    M, N, K = 60800, 8704, 768
    g = torch.Generator(device=DEV).manual_seed(11)
    dy = torch.randn(M, N, device=DEV, generator=g).to(torch.bfloat16)
    h = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    first = _joint_wgrad(dy, h)
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device=DEV)
    for i in range(8):
        if i % 2:
            with torch.cuda.stream(side):
                junk @ junk
        again = _joint_wgrad(dy, h)
        assert torch.equal(first, again), i
    torch.cuda.synchronize()


def test_joint_fc_wgrad_plan_rejects_what_the_kernel_cannot_take():
    import ctypes

    from caiman_asr_amd import _lib

    lib = _lib.lib()
    per = ctypes.c_int64(0)
    assert lib.caiman_joint_fc_wgrad_plan(304000, 8704, 768, _lib.dtype_tag(torch.bfloat16), ctypes.byref(per)) == 5
    assert per.value == 60800
    assert lib.caiman_joint_fc_wgrad_plan(304000, 8704, 700, _lib.dtype_tag(torch.bfloat16), ctypes.byref(per)) == 0
    assert lib.caiman_joint_fc_wgrad_plan(100, 512, 512, _lib.dtype_tag(torch.bfloat16), ctypes.byref(per)) == 0
    assert lib.caiman_joint_fc_wgrad_plan(304000, 8704, 768, _lib.dtype_tag(torch.float32), ctypes.byref(per)) == 0


def test_projection_autograd_takes_the_handwritten_weight_gradient_and_agrees_with_the_library(monkeypatch):
    """train_utils/overlap.py's projection (the joint_fc of the model) at a row count past the library threshold: the weight
    gradient comes from caiman_joint_fc_wgrad by default, and equals the batched library product to fp32 summation order."""
    from caiman_asr_amd.train_utils import overlap

    rows, N, K = 70000, 512, 256
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(rows, K, device=DEV, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV, generator=g) / 16).requires_grad_()
    b = torch.zeros(N, device=DEV, requires_grad=True)
    dy = torch.randn(rows, N, device=DEV, generator=g).to(torch.bfloat16)
    calls, grads = [], []
    real = overlap._joint_wgrad
    monkeypatch.setattr(overlap, "_joint_wgrad", lambda a, c, **kw: (calls.append(1), real(a, c, **kw))[1])
    assert overlap.JOINT_WGRAD, "the hand-written weight gradient is the default"
    for on in (True, False):
        monkeypatch.setattr(overlap, "JOINT_WGRAD", on)
        w.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = overlap.linear_transposed_backward(x, w, b)
        out.backward(dy)
        grads.append(w.grad.clone())
    assert len(calls) == 1
    assert grads[0].dtype == torch.float32
    err = (grads[0] - grads[1]).abs().max().item()
    assert err <= 1e-5 * grads[1].abs().max().item(), err


@pytest.mark.parametrize("P,M,N,K,pad", [
    (1, 259, 256, 256, 0),       # the shortest slice (two pairs of 64-row tiles) and three rows for the library remainder
    (3, 1000, 512, 256, 64),     # operands a constant stride apart (64 spare rows between them), 1000 = 7 x 128 + 104
    (6, 8896, 4096, 1024, 32),   # six post-encoder LSTM layers at B = 32: dR = dG^T . h_prev
    (5, 2300, 1024, 2048, 0),    # K = 2048 (stacked input)
])
def test_batched_weight_gradients_match_fp64_products(P, M, N, K, pad):
    """caiman_wgrad_tn: `P` products of one shape in one launch (the LSTM layers' dW / dR), operands read in place from
    buffers with spare rows between the layers (the pipeline's [layers, T + 1, B, H] activations)."""
    from caiman_asr_amd.train_utils.overlap import wgrad_tn

    g = torch.Generator(device=DEV).manual_seed(P * M + N + K)
    dy_all = torch.randn(P, M + pad, N, device=DEV, generator=g).to(torch.bfloat16)
    x_all = torch.randn(P, M + pad, K, device=DEV, generator=g).to(torch.bfloat16)
    dy, x = dy_all[:, pad:], x_all[:, :M]          # views: different offsets, same constant stride
    x[:, :, 7] = 0.0
    dw = wgrad_tn(dy, x)
    assert dw is not None and dw.shape == (P, N, K) and dw.dtype == torch.float32
    torch.cuda.synchronize()
    for p in range(P):
        ref = dy[p].double().t() @ x[p].double()
        err = (dw[p].double() - ref).abs().max().item()
        assert err <= 2e-5 * ref.abs().max().item(), (p, err)
        assert (dw[p][:, 7] == 0).all()


def test_weight_gradient_kernel_is_deterministic_over_repeated_launches():
    """Race screen for the 8-phase weight-gradient kernel (a transposed read placed before the wait + barrier that retire its
    unit's DMA passes a reference check whenever the DMA happens to land first): the joint projection's product at a quarter of
    the training size, 10 launches, bit-identical results with a second stream varying the timing; checked once against fp64."""
    from caiman_asr_amd.train_utils.overlap import _joint_wgrad

    M, N, K = 76000, 8704, 768
    g = torch.Generator(device=DEV).manual_seed(23)
    dy = torch.randn(M, N, device=DEV, generator=g).to(torch.bfloat16)
    x = torch.randn(M, K, device=DEV, generator=g).to(torch.bfloat16)
    d0 = _joint_wgrad(dy, x)
    assert d0 is not None and d0.dtype == torch.float32
    ref = dy[:, :512].double().t() @ x.double()
    assert (d0[:512].double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV)
    for i in range(10):
        if i % 2:
            with torch.cuda.stream(side):
                for _ in range(4):
                    a @ a
        assert torch.equal(_joint_wgrad(dy, x), d0), i


def test_wgrad_tn_declines_what_it_cannot_address():
    from caiman_asr_amd.train_utils.overlap import wgrad_tn

    a = torch.zeros(2, 512, 256, device=DEV, dtype=torch.bfloat16)
    assert wgrad_tn(a, torch.zeros(2, 512, 240, device=DEV, dtype=torch.bfloat16)) is None      # K = 240 (layer 0)
    assert wgrad_tn(a.transpose(1, 2), a.transpose(1, 2)) is None                                   # rows not contiguous
    assert wgrad_tn(a.float(), a.float()) is None                                                   # fp32 operands
    assert wgrad_tn(a[:, :200], a[:, :200]) is None                                                 # fewer than 256 rows


def test_two_strided_groups_in_one_launch():
    """caiman_wgrad_tn2: the post layers' dR (6 products against [layers, T + 1, B, H] activations from step 0) and dW (5
    products, the same gradients from layer 1 on against the activations from step 1) as ONE launch of 11 products."""
    from caiman_asr_amd.train_utils.overlap import wgrad_tn

    L, T, B, H = 6, 35, 8, 256                      # rows = 280 = 2 x 128 + 24: the library takes the last 24 rows
    g = torch.Generator(device=DEV).manual_seed(17)
    dG = torch.randn(L, T * B, 4 * H, device=DEV, generator=g).to(torch.bfloat16)
    Y = torch.randn(L, T + 1, B, H, device=DEV, generator=g).to(torch.bfloat16)

    def rows3(first, count, skip):
        v = Y[first:first + count, skip:skip + T]
        return torch.as_strided(v, (count, T * B, H), (Y.stride(0), H, 1), v.storage_offset())

    out = wgrad_tn(dG, rows3(0, L, 0), second=(dG[1:], rows3(0, L - 1, 1)))
    assert out is not None and out.shape == (2 * L - 1, 4 * H, H)
    torch.cuda.synchronize()
    for p in range(2 * L - 1):
        dy = dG[p] if p < L else dG[p - L + 1]
        x = Y[p, :T] if p < L else Y[p - L, 1:T + 1]
        ref = dy.double().t() @ x.reshape(T * B, H).double()
        err = (out[p].double() - ref).abs().max().item()
        assert err <= 2e-5 * ref.abs().max().item(), (p, err)
    # a second group of another shape is declined, not mis-read
    assert wgrad_tn(dG, rows3(0, L, 0), second=(dG[1:, :128], rows3(0, L - 1, 1)[:, :128])) is None
