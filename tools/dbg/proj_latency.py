"""K-loop latency of the grouped projection kernel on an almost empty chip: one problem, 8-16 tiles, long K."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.proj_gemm_bench import problem, run, timeit
DEV = "cuda"
dt = torch.bfloat16
for (M, N, K) in ((256, 1024, 4096), (256, 1024, 1024), (512, 4096, 1024), (512, 4096, 4096)):
    a = torch.randn(M, K, device=DEV).to(dt); w = torch.randn(N, K, device=DEV).to(dt); c = torch.empty(M, N, device=DEV, dtype=dt)
    for tile in (2, 5, 8):
        us = timeit(lambda: run([problem(a, w, None, c, M, N, K)], tile), reps=30)
        print(f"M {M} N {N} K {K} tile {tile}: {us:.1f} us  ({us / (K / 64):.2f} us per K step)", flush=True)
    us = timeit(lambda: torch.matmul(a, w.t(), out=c), reps=30)
    print(f"M {M} N {N} K {K} library: {us:.1f} us", flush=True)
