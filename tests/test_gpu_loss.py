"""GPU parity: HIP logsumexp + transducer loss (through the C-ABI) vs the CPU oracle.

Grids follow the reference's own tests (training/lib/tests/logsumexp/test_logsumexp.py,
training/lib/tests/transducer/test_loss.py).  Tolerances: f64 1e-10, f32 2e-5 on O(10) losses,
half types compared against the oracle run on the SAME rounded inputs with 2e-2 abs on grads.
"""
import math

import numpy as np
import pytest
import torch

from tests.helpers import mock_lattice

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _lse_sizes():
    return sorted({v for k in range(15) for v in (2 ** k, 2 ** k + 1, 2 ** k + 3)})


@pytest.mark.parametrize("promote", [True, False])
def test_logsumexp_full_precision(promote):
    from caiman_asr_amd.rnnt_ext.cuda import logsumexp as lse
    from oracle import native

    for n in _lse_sizes():
        x = torch.randn(37, n, dtype=torch.float64, generator=torch.Generator().manual_seed(n))
        out = lse.logsumexp(x.to(DEV), 128, promote)
        assert out.dtype == torch.float64
        assert np.allclose(out.cpu().numpy(), native.logsumexp(x.numpy()), rtol=1e-12), n


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_logsumexp_reduced_precision(dtype):
    from caiman_asr_amd.rnnt_ext.cuda import logsumexp as lse
    from oracle import native

    for n in _lse_sizes() + [8704, 17408]:
        x = torch.randn(29, n, dtype=torch.float64, generator=torch.Generator().manual_seed(n)).to(dtype)
        out = lse.logsumexp(x.to(DEV), 128, True)
        assert out.dtype == torch.float32
        ref = native.logsumexp(x.double().numpy())
        assert np.allclose(out.cpu().numpy(), ref, rtol=2e-6, atol=2e-6), n
        out2 = lse.logsumexp(x.to(DEV), 128, False)
        assert out2.dtype == dtype


def test_logsumexp_strided_rows_and_nonfinite():
    from caiman_asr_amd.rnnt_ext.cuda import logsumexp as lse

    base = torch.randn(11, 300, dtype=torch.float32)
    base[0, 3] = float("nan")
    base[1, 5] = float("inf")
    base[2, :200] = float("-inf")
    x = base.to(DEV)[:, :200]  # row stride 300 > n = 200, misaligned rows
    out = lse.logsumexp(x, 128, True).cpu()
    ref = torch.logsumexp(base[:, :200].double(), -1)
    assert torch.isnan(out[0]) and out[1] == float("inf") and out[2] == float("-inf")
    assert torch.allclose(out[3:].double(), ref[3:], rtol=1e-6)
    with pytest.raises(RuntimeError):
        lse.logsumexp(base.to(DEV).t(), 128, True)
    with pytest.raises(RuntimeError):
        lse.logsumexp(base, 128, True)  # CPU tensor
    assert lse.logsumexp(torch.empty(0, 5, device=DEV), 128, True).shape == (0,)


def _run_gpu(d, dtype, packed, **mods):
    from caiman_asr_amd.rnnt_ext.transducer.loss import TransducerLoss

    x = torch.tensor(d["x_packed"] if packed else d["x_padded"]).to(dtype).to(DEV).requires_grad_(True)
    dbg = []
    loss = TransducerLoss(packed_input=packed)(
        x, torch.tensor(d["label"], device=DEV), torch.tensor(d["f_len"], device=DEV),
        torch.tensor(d["y_len"], device=DEV), d["blank"],
        batch_offset=torch.tensor(d["batch_offset"], device=DEV) if packed else None,
        max_f_len=d["max_f_len"] if packed else None, debug_list=dbg, **mods)
    w = torch.linspace(0.5, 1.5, loss.numel(), device=DEV, dtype=loss.dtype)
    (loss * w).sum().backward()
    return x, loss, dbg, w


def _run_oracle(d, x_np, packed, w, **mods):
    from oracle import native

    kw = dict(delay_penalty=mods.get("delay_penalty", 0.0), eos_penalty=mods.get("eos_penalty", 0.0),
              eos_idx=mods.get("eos_idx"), star_idx=mods.get("star_idx"),
              star_lam=math.log(mods.get("star_penalty", 1.0)))
    extra = dict(batch_offset=d["batch_offset"], max_f_len=d["max_f_len"]) if packed else {}
    a, b, loss, denom = native.transducer_forward(x_np, d["label"], d["f_len"], d["y_len"], d["blank"],
                                                  **extra, **kw)
    g = native.transducer_backward(x_np, denom, w, a, b, d["label"], d["f_len"], d["y_len"], d["blank"],
                                   batch_offset=extra.get("batch_offset"), **kw)
    return a, b, loss, g


TOL = {torch.float64: (1e-10, 1e-10), torch.float32: (3e-5, 2e-5), torch.float16: (2e-3, 2e-3),
       torch.bfloat16: (2e-3, 1e-2)}


@pytest.mark.parametrize("batch_size", [1, 2, 8])
@pytest.mark.parametrize("time_dim", [1, 2, 7])
@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32, torch.float16, torch.bfloat16])
def test_loss_and_grad_match_oracle(batch_size, time_dim, packed, dtype):
    d = mock_lattice(batch_size, time_dim, seed=batch_size * 13 + time_dim)
    x, loss, dbg, w = _run_gpu(d, dtype, packed)
    x_np = x.detach().double().cpu().numpy()
    a, b, ref_loss, ref_g = _run_oracle(d, x_np, packed, w.double().cpu().numpy())
    ltol, gtol = TOL[dtype]
    assert np.allclose(loss.detach().cpu().double().numpy(), ref_loss, rtol=ltol, atol=ltol)
    assert np.allclose(x.grad.double().cpu().numpy(), ref_g, atol=gtol)
    # alpha / beta on the valid region
    ga, gb = dbg[0].double().cpu().numpy(), dbg[1].double().cpu().numpy()
    for i in range(batch_size):
        T, U = d["f_len"][i], d["y_len"][i] + 1
        assert np.allclose(ga[i, :T, :U], a[i, :T, :U], rtol=ltol, atol=ltol * 10)
        assert np.allclose(gb[i, :T, :U], b[i, :T, :U], rtol=ltol, atol=ltol * 10)


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("delay_penalty", [0.0, 0.05, 0.99, 2.0])
@pytest.mark.parametrize("eos", [(None, 0.0), (1, 0.1), (1, 0.5)])
@pytest.mark.parametrize("star", [(None, 1.0), (2, 0.1), (2, 0.5)])
def test_modifier_grid_f64(packed, delay_penalty, eos, star):
    # training/lib/tests/transducer/test_loss.py:208-260 (gradcheck grid); here the oracle's
    # analytic gradient (itself pinned by finite differences on CPU) is the checker.
    eos_idx, eos_penalty = eos
    star_idx, star_penalty = star
    d = mock_lattice(4, 7, seed=17, eos_idx=eos_idx, star_idx=star_idx)
    mods = dict(delay_penalty=delay_penalty, eos_penalty=eos_penalty, eos_idx=eos_idx, star_idx=star_idx,
                star_penalty=star_penalty)
    x, loss, _, w = _run_gpu(d, torch.float64, packed, **mods)
    _, _, ref_loss, ref_g = _run_oracle(d, x.detach().cpu().numpy(), packed, w.cpu().numpy(), **mods)
    assert np.allclose(loss.detach().cpu().numpy(), ref_loss, rtol=1e-10)
    assert np.allclose(x.grad.cpu().numpy(), ref_g, atol=1e-10)


def test_pack_no_pack_equivalent_on_gpu():
    d = mock_lattice(8, 7, seed=23)
    _, l0, _, _ = _run_gpu(d, torch.float32, False, delay_penalty=0.05)
    _, l1, _, _ = _run_gpu(d, torch.float32, True, delay_penalty=0.05)
    assert torch.allclose(l0, l1, rtol=1e-6)


def test_long_lattice_multi_wave_and_big_vocab():
    # U+1 > 64 exercises the cross-wave LDS hand-off; V = 8704 the 16-byte vector path.
    rng = np.random.default_rng(5)
    B, T, U, V = 2, 40, 130, 8704
    d = dict(label=rng.integers(0, V - 1, size=(B, U)).astype(np.int32),
             f_len=np.array([T, T - 7], dtype=np.int32), y_len=np.array([U, 70], dtype=np.int32), blank=V - 1)
    d["batch_offset"] = np.cumsum(d["f_len"].astype(np.int64) * (d["y_len"] + 1))
    d["max_f_len"] = T
    rows = int(d["batch_offset"][-1])
    d["x_packed"] = rng.standard_normal((rows, V)).astype(np.float32)
    x, loss, _, w = _run_gpu(d, torch.bfloat16, True, delay_penalty=0.01)
    x_np = x.detach().double().cpu().numpy()
    _, _, ref_loss, ref_g = _run_oracle(d, x_np, True, w.double().cpu().numpy(), delay_penalty=0.01)
    assert np.allclose(loss.detach().cpu().numpy(), ref_loss, rtol=1e-4)
    # gradients are O(1e-4..1) and stored in bf16: compare relative to bf16 resolution
    g = x.grad.double().cpu().numpy()
    assert np.allclose(g, ref_g, atol=4e-3, rtol=1.6e-2)


def test_nan_denominator_propagates_and_errors():
    from caiman_asr_amd.rnnt_ext.cuda import transducer_loss as tl
    from caiman_asr_amd.rnnt_ext.transducer.loss import TransducerLoss

    d = mock_lattice(2, 3, seed=7, full=True)
    x = torch.tensor(d["x_padded"], dtype=torch.float32, device=DEV)
    x[0, 0, 0, 0] = float("inf")
    loss = TransducerLoss()(x, torch.tensor(d["label"], device=DEV), torch.tensor(d["f_len"], device=DEV),
                            torch.tensor(d["y_len"], device=DEV), d["blank"])
    assert torch.isnan(loss[0]) and torch.isfinite(loss[1])
    with pytest.raises(RuntimeError, match="Expected blank index"):
        TransducerLoss()(x, torch.tensor(d["label"], device=DEV), torch.tensor(d["f_len"], device=DEV),
                         torch.tensor(d["y_len"], device=DEV), 999)
    with pytest.raises(RuntimeError, match="contiguous"):
        tl.forward(x.transpose(1, 2), x[..., 0].contiguous(), torch.tensor(d["label"], device=DEV),
                   torch.tensor(d["f_len"], device=DEV), torch.tensor(d["y_len"], device=DEV), torch.empty(0),
                   0.0, 3, d["blank"], 0.0, -1, 0.0, -2, False)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("V,B,T,U", [(29, 3, 9, 5), (8704, 2, 41, 6), (12288, 2, 23, 3), (17408, 2, 19, 4), (40960, 1, 9, 3)])
def test_backward_with_fused_column_sums(dtype, packed, V, B, T, U):
    """caiman_transducer_loss_backward_colsum: the same gradient as the plain backward, bit for bit, plus its column
    sums (the bias gradient of the projection that produced the logits; the reference takes `grad_output.sum(0)` in a
    separate pass).  Row counts that are no multiple of the 64 rows a workgroup takes.  V <= 10 240 (16-bit): four waves per
    workgroup; 12 288 and 17 408: eight waves at 3 and 5 chunks per lane; 40 960: eight waves, 10 chunks (the 16-chunk build);
    fp32 doubles the chunk counts (17 408: 9 per lane of eight waves)."""
    import caiman_asr_amd.rnnt_ext.cuda.logsumexp as lse
    import caiman_asr_amd.rnnt_ext.cuda.transducer_loss as tl
    from tests.helpers import mock_lattice

    d = mock_lattice(B, T, vocab=V, max_decode_length=U + 1, seed=V + B, packed=packed)
    x = torch.tensor(d["x"], device=DEV).to(dtype).contiguous()
    label = torch.tensor(d["label"], device=DEV)
    f_len, y_len = torch.tensor(d["f_len"], device=DEV), torch.tensor(d["y_len"], device=DEV)
    bo = torch.tensor(d["batch_offset"], device=DEV) if packed else torch.empty(0, device=DEV, dtype=torch.int64)
    denom = lse.logsumexp(x.view(-1, V), 128, True).view(x.shape[:-1])
    args = (0.01, d["max_f_len"] if packed else T, d["blank"], 0.0, -1, 0.0, -2, packed)
    alpha, beta, loss = tl.forward(x, denom, label, f_len, y_len, bo, *args)
    lg = torch.full((B,), 1.0 / B, device=DEV)
    ref = tl.backward(x, denom, lg, alpha, beta, f_len, y_len, label, bo, *args)
    got, colsum = tl.backward_colsum(x, denom, lg, alpha, beta, f_len, y_len, label, bo, *args)
    assert torch.equal(got, ref)
    want = ref.view(-1, V).double().sum(0)
    tol = 1e-6 if dtype == torch.float32 else 1e-5     # fp32 sums of rounded values, a different order than torch's
    assert torch.allclose(colsum.double(), want, atol=tol * max(1.0, float(want.abs().max())) + 1e-7)


def test_projection_backward_picks_up_the_fused_bias_gradient():
    """The joint projection's backward (train_utils/overlap.py) takes its bias gradient from the loss backward kernel when
    the gradient tensor it receives is the one that kernel wrote, and falls back to a reduction otherwise."""
    import caiman_asr_amd.rnnt_ext.transducer.loss as L
    from caiman_asr_amd.train_utils.overlap import linear_transposed_backward
    from tests.helpers import mock_lattice

    V, H = 64, 32
    d = mock_lattice(3, 7, vocab=V, max_decode_length=5, seed=9, packed=True)
    rows = d["x"].shape[0]
    torch.manual_seed(0)
    h = torch.randn(rows, H, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    label = torch.tensor(d["label"], device=DEV)
    f_len, y_len = torch.tensor(d["f_len"], device=DEV), torch.tensor(d["y_len"], device=DEV)
    bo = torch.tensor(d["batch_offset"], device=DEV)
    grads = []
    for fuse in (True, False):
        L.FUSE_BIAS_GRADIENT = fuse
        try:
            torch.manual_seed(5)
            w = torch.nn.Parameter(torch.randn(V, H, device=DEV) * 0.2)
            b = torch.nn.Parameter(torch.zeros(V, device=DEV))
            with torch.autocast("cuda", dtype=torch.bfloat16):
                logits = linear_transposed_backward(h, w, b)
                loss = L.TransducerLoss(packed_input=True)(logits, label, f_len, y_len, d["blank"], batch_offset=bo,
                                                           max_f_len=d["max_f_len"]).mean()
            loss.backward()
            assert L._latest_colsum is None          # consumed (or never produced)
            grads.append(b.grad.clone())
        finally:
            L.FUSE_BIAS_GRADIENT = True
    assert torch.allclose(grads[0], grads[1], atol=2e-3 * float(grads[1].abs().max()) + 1e-6)
