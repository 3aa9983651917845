"""Pin the CPU oracle (oracle/rnnt_oracle.c) before anything is compared with it.

The reference has no CPU transducer loss / logsumexp and its CUDA kernels cannot be
built here, and its own tests hold no known-answer constants for them,
so the pins are: (1) the path-enumeration definition of the loss, (2) the same
differential / self-consistency properties the reference's tests assert
(training/lib/tests/transducer/test_loss.py, .../logsumexp/test_logsumexp.py,
.../custom_lstm/test_cuda.py), (3) finite-difference gradients.
"""
import math

import numpy as np
import pytest
import torch

from oracle import brute, native
from tests.helpers import mock_lattice, unpack


def _log_softmax(x):
    m = x.max(-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))


@pytest.mark.parametrize("T,U", [(1, 0), (1, 3), (3, 1), (2, 2), (4, 3), (3, 4)])
@pytest.mark.parametrize("mods", [
    dict(), dict(delay_penalty=0.05), dict(delay_penalty=2.0),
    dict(eos_penalty=0.5, eos_idx=1, delay_penalty=0.1),
    dict(star_idx=2, star_lam=math.log(0.5)),
    dict(star_idx=2, star_lam=math.log(0.1), eos_idx=1, eos_penalty=0.1, delay_penalty=0.99),
])
def test_loss_matches_path_enumeration(T, U, mods):
    rng = np.random.default_rng(T * 10 + U)
    V, blank = 6, 5
    x = rng.standard_normal((1, T, U + 1, V))
    label = rng.integers(0, 4, size=(1, max(U, 1))).astype(np.int32)
    if U > 0 and "eos_idx" in mods:
        label[0, U - 1] = mods["eos_idx"]
    if U > 1 and "star_idx" in mods:
        label[0, 0] = mods["star_idx"]
    # label tensor must be [B, Umax]; pad x to Umax+1 rows when U == 0
    Umax = label.shape[1]
    if Umax + 1 > U + 1:
        x = np.concatenate([x, rng.standard_normal((1, T, Umax - U, V))], 2)
    _, _, loss, _ = native.transducer_forward(x, label, [T], [U], blank, **mods)
    lp = _log_softmax(x[0])
    ref = brute.loss_by_enumeration(
        lp, list(label[0, :U]), T, U, blank,
        delay_penalty=mods.get("delay_penalty", 0.0), eos_penalty=mods.get("eos_penalty", 0.0),
        eos_idx=mods.get("eos_idx", -1), star_lam=mods.get("star_lam", 0.0),
        star_idx=mods.get("star_idx", -2))
    assert np.allclose(loss[0], ref, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("batch_size", [2, 8])
@pytest.mark.parametrize("time_dim", [1, 2, 7])
@pytest.mark.parametrize("delay_penalty", [0.0, 0.05])
@pytest.mark.parametrize("eos", [(None, 0.0), (1, 0.1), (1, 0.5)])
@pytest.mark.parametrize("star_idx", [None, 2])
def test_pack_no_pack_equivalent(batch_size, time_dim, delay_penalty, eos, star_idx):
    # training/lib/tests/transducer/test_loss.py:65-124
    eos_idx, eos_penalty = eos
    d = mock_lattice(batch_size, time_dim, seed=batch_size * 31 + time_dim)
    kw = dict(delay_penalty=delay_penalty, eos_penalty=eos_penalty, eos_idx=eos_idx,
              star_idx=star_idx, star_lam=math.log(0.75))
    _, _, l_pad, _ = native.transducer_forward(d["x_padded"], d["label"], d["f_len"], d["y_len"],
                                               d["blank"], **kw)
    _, _, l_pack, _ = native.transducer_forward(d["x_packed"], d["label"], d["f_len"], d["y_len"],
                                                d["blank"], batch_offset=d["batch_offset"],
                                                max_f_len=d["max_f_len"], **kw)
    assert np.allclose(l_pad, l_pack, rtol=1e-13)


def test_alpha_beta_consistency():
    # -alpha-side loss equals -beta[0,0]: alpha(T-1,U) + null(T-1,U) = beta(0,0)
    d = mock_lattice(4, 6, seed=3)
    a, b, loss, denom = native.transducer_forward(d["x_padded"], d["label"], d["f_len"], d["y_len"],
                                                  d["blank"], delay_penalty=0.3)
    for i in range(4):
        T, U = d["f_len"][i], d["y_len"][i]
        lp_blank = d["x_padded"][i, T - 1, U, d["blank"]] - denom[i, T - 1, U]
        assert np.isclose(a[i, T - 1, U] + lp_blank, -loss[i], rtol=1e-12)
        # and every cell: alpha+beta marginal never exceeds the total
        tot = a[i, :T, : U + 1] + b[i, :T, : U + 1]
        assert np.all(tot <= -loss[i] + 1e-9)


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("delay_penalty", [0.0, 0.05, 0.99, 2.0])
@pytest.mark.parametrize("eos", [(None, 0.0), (1, 0.1), (1, 0.5)])
@pytest.mark.parametrize("star", [(None, 1.0), (2, 0.1), (2, 0.5)])
def test_backward_matches_finite_differences(packed, delay_penalty, eos, star):
    # the reference pins its backward with gradcheck over this grid
    # (training/lib/tests/transducer/test_loss.py:208-260)
    eos_idx, eos_penalty = eos
    star_idx, star_penalty = star
    d = mock_lattice(2, 4, vocab=6, max_decode_length=4, seed=11, packed=packed,
                     eos_idx=eos_idx, star_idx=star_idx)
    kw = dict(delay_penalty=delay_penalty, eos_penalty=eos_penalty, eos_idx=eos_idx,
              star_idx=star_idx, star_lam=math.log(star_penalty))
    extra = dict(batch_offset=d["batch_offset"], max_f_len=d["max_f_len"]) if packed else {}
    x = d["x"].copy()
    w = np.array([0.7, 1.3])  # upstream gradient (must be positive: log(loss_grad))

    def f(xx):
        return float((native.transducer_forward(xx, d["label"], d["f_len"], d["y_len"], d["blank"],
                                                **extra, **kw)[2] * w).sum())

    a, b, _, denom = native.transducer_forward(x, d["label"], d["f_len"], d["y_len"], d["blank"],
                                               **extra, **kw)
    g = native.transducer_backward(x, denom, w, a, b, d["label"], d["f_len"], d["y_len"], d["blank"],
                                   batch_offset=extra.get("batch_offset"), **kw)
    num = np.zeros_like(x)
    eps = 1e-6
    it = np.nditer(x, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        xp = x.copy(); xp[i] += eps
        xm = x.copy(); xm[i] -= eps
        num[i] = (f(xp) - f(xm)) / (2 * eps)
    assert np.allclose(g, num, atol=2e-7, rtol=1e-5)


def test_backward_padded_region_is_zero():
    d = mock_lattice(3, 5, seed=5)
    a, b, _, denom = native.transducer_forward(d["x_padded"], d["label"], d["f_len"], d["y_len"],
                                               d["blank"])
    g = native.transducer_backward(d["x_padded"], denom, np.ones(3), a, b, d["label"], d["f_len"],
                                   d["y_len"], d["blank"])
    for i in range(3):
        T, U = d["f_len"][i], d["y_len"][i]
        assert np.all(g[i, T:] == 0) and np.all(g[i, :, U + 1:] == 0)
        assert np.any(g[i, :T, : U + 1] != 0)


def test_nonfinite_denominator_gives_nan_loss():
    # sub_or_nan: training/lib/csrc/transducer_loss.cu:59-62
    d = mock_lattice(2, 3, seed=7, full=True)
    x = d["x_padded"].copy()
    x[0, 0, 0, 0] = np.inf
    _, _, loss, _ = native.transducer_forward(x, d["label"], d["f_len"], d["y_len"], d["blank"])
    assert np.isnan(loss[0]) and np.isfinite(loss[1])


@pytest.mark.parametrize("n", sorted({v for k in range(13) for v in (2 ** k, 2 ** k + 1, 2 ** k + 3)}))
def test_logsumexp_matches_torch(n):
    # training/lib/tests/logsumexp/test_logsumexp.py:29-44
    x = np.random.default_rng(n).standard_normal((50, n))
    assert np.allclose(native.logsumexp(x), torch.logsumexp(torch.from_numpy(x), -1).numpy(),
                       rtol=1e-13)


def test_logsumexp_nonfinite_rows():
    x = np.zeros((4, 5))
    x[0, 1] = np.nan
    x[1, 2] = np.inf
    x[2, :] = -np.inf
    out = native.logsumexp(x)
    assert np.isnan(out[0]) and out[1] == np.inf and out[2] == -np.inf
    assert np.isclose(out[3], math.log(5))


@pytest.mark.parametrize("T,B,I,H", [(1, 1, 1, 1), (7, 3, 2, 5), (8, 4, 16, 16)])
def test_lstm_soft_matches_torch_lstm(T, B, I, H):
    # the reference pins Kind::soft by equality with torch.nn.LSTM
    # (training/lib/tests/custom_lstm/test_cuda.py:137-217)
    torch.manual_seed(T * 100 + H)
    ref = torch.nn.LSTM(I, H, 1).double()
    x = torch.randn(T, B, I, dtype=torch.float64, requires_grad=True)
    h0 = torch.randn(1, B, H, dtype=torch.float64)
    c0 = torch.randn(1, B, H, dtype=torch.float64)
    out, (hn, cn) = ref(x, (h0, c0))
    W, R = ref.weight_ih_l0.detach().numpy(), ref.weight_hh_l0.detach().numpy()
    bias = (ref.bias_ih_l0 + ref.bias_hh_l0).detach().numpy()
    gates = x.detach().numpy() @ W.T + bias
    g, c, y = native.lstm_fwd(R, gates, c0[0].numpy(), h0[0].numpy())
    assert np.allclose(y[1:], out.detach().numpy(), atol=1e-12)
    assert np.allclose(c[-1], cn[0].detach().numpy(), atol=1e-12)
    # backward vs autograd
    delta = torch.randn_like(out)
    out.backward(delta)
    dG, _ = native.lstm_bwd(R, g, c, delta.numpy())
    dG2 = dG.reshape(T * B, 4 * H)
    assert np.allclose(dG2 @ W, x.grad.numpy().reshape(T * B, I), atol=1e-11)
    assert np.allclose(dG2.T @ x.detach().numpy().reshape(T * B, I), ref.weight_ih_l0.grad.numpy(), atol=1e-11)
    assert np.allclose(dG2.T @ y[:-1].reshape(T * B, H), ref.weight_hh_l0.grad.numpy(), atol=1e-11)
    assert np.allclose(dG2.sum(0), ref.bias_ih_l0.grad.numpy(), atol=1e-11)


@pytest.mark.parametrize("T,B,H", [(1, 1, 1), (7, 3, 5)])
def test_lstm_hard_gradients_by_finite_differences(T, B, H):
    # Kind::hard (training/lib/csrc/lstm.cu:41-76); the reference uses gradcheck
    # (training/lib/tests/custom_lstm/test_cuda.py:12-42).
    rng = np.random.default_rng(T + H)
    R = rng.standard_normal((4 * H, H)) * 0.5
    gates = rng.standard_normal((T, B, 4 * H))
    c0 = rng.standard_normal((B, H)) * 0.3
    y0 = rng.standard_normal((B, H)) * 0.3
    w = rng.standard_normal((T, B, H))

    def f(gg):
        return float((native.lstm_fwd(R, gg, c0, y0, hard=True)[2][1:] * w).sum())

    g, c, y = native.lstm_fwd(R, gates, c0, y0, hard=True)
    dG, _ = native.lstm_bwd(R, g, c, w, hard=True)
    eps = 1e-7
    num = np.zeros_like(gates)
    it = np.nditer(gates, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        gp = gates.copy(); gp[i] += eps
        gm = gates.copy(); gm[i] -= eps
        num[i] = (f(gp) - f(gm)) / (2 * eps)
    # dG is the gradient w.r.t. the PRE-activation gates
    assert np.allclose(dG, num, atol=1e-5)


def test_frontend_oracle_reproduces_the_reference_golden_logmel():
    """The reference pins its DALI frontend with tests/test_data/audio_tensor_batch.pt at atol 2e-4
    (training/tests/data/dali/test_data_loader.py:235-258).  The numpy oracle reproduces that tensor from
    the decoded recording: mean |err| ~1.6e-6, max 2.2e-4 (the reference tensor was produced in float32 with
    1e-5 dither, hence the slightly wider bound on the max)."""
    import os

    from oracle import frontend as of

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "frontend_ref.npz"))
    x = g["pcm"].astype(np.float64) / 32768.0
    lm = of.logmel(x, win=int(g["window_size"] * 16000), hop=160, nfft=512, nmel=80, initial_pad=0)
    assert lm.shape == g["logmel_norm"].shape == (80, 888)
    nm = of.normalize(lm, lm.shape[1])
    err = np.abs(nm - g["logmel_norm"])
    assert err.max() < 3e-4 and err.mean() < 5e-6
    assert (err > 2e-4).sum() <= 3
