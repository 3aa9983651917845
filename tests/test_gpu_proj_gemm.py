"""Grouped input-projection GEMM (csrc/proj_gemm.hip, C-ABI caiman_proj_gemm) against a plain PyTorch fp32 product of the
same 16-bit operands -- the op it replaces is `torch.addmm(bias, x, W_ih.t())` per layer
(training/lib/src/rnnt_ext/custom_lstm/lstm.py:51-55) and its input gradient `dG @ W_ih`.

Tolerance: fp32 accumulation of exact 16-bit products, one rounding of the result to the storage type: half an ulp of
the largest magnitude (2^-8 relative for bf16 -- 7 stored mantissa bits --, 2^-11 for f16) plus the fp32 summation-order noise (K * 2^-24 relative).
"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _problem(a_ptr, w, bias, c_ptr, M, N, K, a_rows, c_rows):
    from caiman_asr_amd import _lib

    return _lib.ProjProblem(a_ptr, w.data_ptr(), bias.data_ptr() if bias is not None else None, c_ptr, M, N, K,
                            a_rows[0], a_rows[4], c_rows[0], c_rows[4], a_rows[1], a_rows[2], a_rows[3],
                            c_rows[1], c_rows[2], c_rows[3])


def _plain(t, M, W):
    return (M, 0, t.stride(0), 0, W)


def _run(problems, dt, tile):
    from caiman_asr_amd import _lib

    arr = (_lib.ProjProblem * len(problems))(*problems)
    _lib.check(_lib.lib().caiman_proj_gemm(ctypes.cast(arr, ctypes.c_void_p), len(problems), _lib.dtype_tag(dt), tile,
                                           _lib.stream()))
    torch.cuda.synchronize()


def _tol(ref, dt, K):
    half_ulp = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    return (half_ulp + K * 2.0 ** -24) * float(ref.abs().max()) + 1e-6


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 8, 9, 10])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,bias", [(512, 4096, 1024, True), (224, 512, 256, True), (37, 256, 128, False),
                                        (1, 128, 128, True), (300, 1024, 4096, False),
                                        # K / 64 = 2 .. 7 steps: every entry and exit of the three-stage pipeline
                                        (70, 128, 256, True), (70, 128, 384, False), (70, 128, 512, True), (70, 128, 640, False),
                                        (70, 128, 768, True), (70, 128, 896, False)])
def test_single_problem_matches_fp32_product(M, N, K, bias, dt, tile):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dt).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).to(DEV)
    b = torch.randn(N, generator=g).to(dt).to(DEV) if bias else None
    c = torch.full((M + 1, N), 7.0, dtype=dt, device=DEV)      # one guard row behind the output
    _run([_problem(a.data_ptr(), w, b, c.data_ptr(), M, N, K, _plain(a, M, K), _plain(c, M, N))], dt, tile)
    ref = a.float() @ w.float().t() + (b.float() if bias else 0.0)
    assert float((c[:M].float() - ref).abs().max()) <= _tol(ref, dt, K)
    assert bool((c[M] == 7.0).all()), "rows past M must not be written"


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 8, 9, 10])
def test_grouped_launch_with_stacked_views(tile):
    """Several problems of different K in one launch, among them the StackTime read (A rows gather f frames) and the
    StackTime scatter (C columns go to f frames): what one pipeline tick of encoder_pipe.py issues."""
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(11)
    B, H, f, T2 = 24, 256, 2, 5          # B not a multiple of 32, ragged last tile
    src = torch.randn(T2 * f, B, H, generator=g).to(dt).to(DEV)
    w_st = (torch.randn(4 * H, f * H, generator=g) / (f * H) ** 0.5).to(dt).to(DEV)
    b_st = torch.randn(4 * H, generator=g).to(dt).to(DEV)
    c_st = torch.zeros(T2 * B, 4 * H, dtype=dt, device=DEV)
    x2 = torch.randn(3 * B, H, generator=g).to(dt).to(DEV)
    w2 = (torch.randn(4 * H, H, generator=g) / H ** 0.5).to(dt).to(DEV)
    c2 = torch.zeros(3 * B, 4 * H, dtype=dt, device=DEV)
    dg = torch.randn(T2 * B, 4 * H, generator=g).to(dt).to(DEV)
    wt = (torch.randn(f * H, 4 * H, generator=g) / (4 * H) ** 0.5).to(dt).to(DEV)
    out = torch.zeros(T2 * f, B, H, dtype=dt, device=DEV)
    probs = [
        _problem(dg.data_ptr(), wt, None, out.data_ptr(), T2 * B, f * H, 4 * H, _plain(dg, T2 * B, 4 * H),
                 (B, f * B * H, H, B * H, H)),
        _problem(src.data_ptr(), w_st, b_st, c_st.data_ptr(), T2 * B, 4 * H, f * H, (B, f * B * H, H, B * H, H),
                 _plain(c_st, T2 * B, 4 * H)),
        _problem(x2.data_ptr(), w2, None, c2.data_ptr(), 3 * B, 4 * H, H, _plain(x2, 3 * B, H), _plain(c2, 3 * B, 4 * H)),
    ]
    _run(probs, dt, tile)
    ref = src.view(T2, f, B, H).transpose(1, 2).reshape(T2 * B, f * H).float() @ w_st.float().t() + b_st.float()
    assert float((c_st.float() - ref).abs().max()) <= _tol(ref, dt, f * H)
    ref = x2.float() @ w2.float().t()
    assert float((c2.float() - ref).abs().max()) <= _tol(ref, dt, H)
    ref = (dg.float() @ wt.float().t()).view(T2, B, f, H).transpose(1, 2).reshape(T2 * f, B, H)
    assert float((out.float() - ref).abs().max()) <= _tol(ref, dt, 4 * H)


def test_geometry_outside_the_kernel_is_refused_not_computed():
    from caiman_asr_amd import _lib

    dt = torch.bfloat16
    a = torch.zeros(64, 96, dtype=dt, device=DEV)
    w = torch.zeros(128, 96, dtype=dt, device=DEV)
    c = torch.zeros(64, 128, dtype=dt, device=DEV)
    p = _problem(a.data_ptr(), w, None, c.data_ptr(), 64, 128, 96, _plain(a, 64, 96), _plain(c, 64, 128))   # K % 128 != 0
    assert _lib.lib().caiman_proj_gemm_supported(ctypes.byref(p), _lib.dtype_tag(dt)) == 0
    arr = (_lib.ProjProblem * 1)(p)
    with pytest.raises(RuntimeError):
        _lib.check(_lib.lib().caiman_proj_gemm(ctypes.cast(arr, ctypes.c_void_p), 1, _lib.dtype_tag(dt), 0, _lib.stream()))
    a32 = torch.zeros(128, 128, device=DEV)
    p = _problem(a32.data_ptr(), a32, None, a32.data_ptr(), 128, 128, 128, _plain(a32, 128, 128), _plain(a32, 128, 128))
    assert _lib.lib().caiman_proj_gemm_supported(ctypes.byref(p), _lib.dtype_tag(torch.float32)) == 0
