"""Grouped input-projection GEMM (csrc/proj_gemm.hip) against the library calls it replaces: agreement with an fp32
product of the same bf16 operands, and time per pipeline tick at the shapes of the base encoder (B = 32).

    python tools/proj_gemm_bench.py [--tile 0|1|2]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caiman_asr_amd import _lib  # noqa: E402

DEV = "cuda"


def problem(a, w, bias, c, M, N, K, a_rows=None, c_rows=None):
    """plain row-major problem; a_rows / c_rows = (inner, stride_outer, stride_inner, stride_seg, seg_len) or None."""
    ai = a_rows or (M, 0, a.stride(0), 0, K)
    ci = c_rows or (M, 0, c.stride(0), 0, N)
    return _lib.ProjProblem(a.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, c.data_ptr(), M, N, K,
                            ai[0], ai[4], ci[0], ci[4], ai[1], ai[2], ai[3], ci[1], ci[2], ci[3])


def run(problems, tile):
    lib = _lib.lib()
    arr = (_lib.ProjProblem * len(problems))(*problems)
    _lib.check(lib.caiman_proj_gemm(ctypes.cast(arr, ctypes.c_void_p), len(problems), _lib.dtype_tag(torch.bfloat16), tile,
                                    _lib.stream()))


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tile", type=int, default=0)
    args = ap.parse_args()
    torch.manual_seed(0)
    dt = torch.bfloat16
    H, B = 1024, 32

    def mk(M, N, K, bias=True):
        a = torch.randn(M, K, device=DEV).to(dt)
        w = (torch.randn(N, K, device=DEV) / K ** 0.5).to(dt)
        b = torch.randn(N, device=DEV).to(dt) if bias else None
        c = torch.empty(M, N, device=DEV, dtype=dt)
        return a, w, b, c

    # ---- agreement --------------------------------------------------------------------------------------------
    for tile in (2, 5, 8, 9, 10):
        for (M, N, K) in ((512, 4096, 1024), (1024, 4096, 1024), (224, 4096, 2048), (512, 1024, 4096), (37, 256, 128), (300, 256, 256), (64, 128, 384), (64, 128, 640)):
            a, w, b, c = mk(M, N, K)
            run([problem(a, w, b, c, M, N, K)], tile)
            ref = a.float() @ w.float().t() + b.float()
            err = (c.float() - ref).abs().max().item()
            scale = ref.abs().max().item()
            print(f"tile {tile}  [{M} x {K}] x [{N} x {K}]^T: max err {err:.4f} (|ref| max {scale:.1f})", flush=True)
            assert err <= 2.0 ** -8 * scale + 1e-3, "projection GEMM disagrees with the fp32 product"
    # StackTime view as A: src [T, B, H] -> rows (t2, b) of 2H features
    T2, f = 16, 2
    src = torch.randn(T2 * f, B, H, device=DEV).to(dt)
    w = (torch.randn(4 * H, f * H, device=DEV) / (f * H) ** 0.5).to(dt)
    c = torch.empty(T2 * B, 4 * H, device=DEV, dtype=dt)
    run([problem(src, w, None, c, T2 * B, 4 * H, f * H, a_rows=(B, f * B * H, H, B * H, H))], 0)
    ref = src.view(T2, f, B, H).transpose(1, 2).reshape(T2 * B, f * H).float() @ w.float().t()
    print("stacked A: max err", (c.float() - ref).abs().max().item(), flush=True)
    assert (c.float() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-3
    # and as C (the scatter of the input gradient through StackTime)
    dg = torch.randn(T2 * B, 4 * H, device=DEV).to(dt)
    wt = (torch.randn(f * H, 4 * H, device=DEV) / (4 * H) ** 0.5).to(dt)
    out = torch.zeros(T2 * f, B, H, device=DEV, dtype=dt)
    run([problem(dg, wt, None, out, T2 * B, f * H, 4 * H, c_rows=(B, f * B * H, H, B * H, H))], 0)
    ref = (dg.float() @ wt.float().t()).view(T2, B, f, H).transpose(1, 2).reshape(T2 * f, B, H)
    print("stacked C: max err", (out.float() - ref).abs().max().item(), flush=True)
    assert (out.float() - ref).abs().max().item() <= 2.0 ** -8 * ref.abs().max().item() + 1e-3

    # ---- one forward tick of the base encoder: pre layer 1 (1024 rows), post layer 0 (512 rows, K = 2H), post 1-4 -----
    fw = [mk(512, 4 * H, 2 * H), mk(1024, 4 * H, H)] + [mk(512, 4 * H, H) for _ in range(4)]
    probs = [problem(a, w, b, c, a.shape[0], w.shape[0], a.shape[1]) for a, w, b, c in fw]
    flops = sum(2 * a.shape[0] * w.shape[0] * a.shape[1] for a, w, b, c in fw)
    wts = [w.t().contiguous() for a, w, b, c in fw]     # [K, N]: the layout the library path uses in the forward pass

    def lib_tick():
        for (a, w, b, c), wt_ in zip(fw, wts):
            torch.addmm(b, a, wt_, out=c)

    Xb = torch.stack([fw[i][0] for i in range(2, 6)])
    Wb = torch.stack([wts[i] for i in range(2, 6)])
    bb = torch.stack([fw[i][2] for i in range(2, 6)]).unsqueeze(1)
    Ob = torch.empty(4, 512, 4 * H, device=DEV, dtype=dt)

    def lib_tick_bmm():
        torch.addmm(fw[0][2], fw[0][0], wts[0], out=fw[0][3])
        torch.addmm(fw[1][2], fw[1][0], wts[1], out=fw[1][3])
        torch.baddbmm(bb, Xb, Wb, out=Ob)

    for tile in ((args.tile,) if args.tile else (2, 5, 8, 9, 10)):
        us = timeit(lambda: run(probs, tile))
        print(f"forward tick, grouped kernel tile {tile}: {us:.1f} us = {flops / us / 1e6:.0f} TFLOP/s", flush=True)
    us = timeit(lib_tick)
    print(f"forward tick, 6 library calls: {us:.1f} us = {flops / us / 1e6:.0f} TFLOP/s", flush=True)
    us = timeit(lib_tick_bmm)
    print(f"forward tick, 2 calls + one batched call (what the pipeline does): {us:.1f} us = {flops / us / 1e6:.0f} TFLOP/s", flush=True)

    # ---- one backward tick: delta = dG @ W (N = H or 2H, K = 4H) ----------------------------------------------------
    bw = [mk(512, 2 * H, 4 * H, bias=False), mk(1024, H, 4 * H, bias=False)] + [mk(512, H, 4 * H, bias=False) for _ in range(4)]
    probs = [problem(a, w, None, c, a.shape[0], w.shape[0], a.shape[1]) for a, w, b, c in bw]
    flops = sum(2 * a.shape[0] * w.shape[0] * a.shape[1] for a, w, b, c in bw)

    def lib_tick_b():
        for a, w, b, c in bw:
            torch.matmul(a, w.t(), out=c)

    for tile in ((args.tile,) if args.tile else (2, 5, 8, 9, 10)):
        us = timeit(lambda: run(probs, tile))
        print(f"backward tick, grouped kernel tile {tile}: {us:.1f} us = {flops / us / 1e6:.0f} TFLOP/s", flush=True)
    us = timeit(lib_tick_b)
    print(f"backward tick, 6 library calls: {us:.1f} us = {flops / us / 1e6:.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
