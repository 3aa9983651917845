"""Layouts for the LSTM pipeline's per-tick GEMMs (B = 32, chunk 32 / 16, H = 1024), bf16."""
import time

import torch

dev = "cuda"


def bench(name, fn, flops):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print(f"{name:64s} {dt * 1e6:8.1f} us  {flops / dt / 1e12:7.1f} TFLOP/s")


H = 1024
for rows, cnt in ((512, 5), (1024, 1)):
    X = torch.randn(cnt, rows, H, device=dev, dtype=torch.bfloat16)
    W = torch.randn(cnt, 4 * H, H, device=dev, dtype=torch.bfloat16)       # [4H, K] as stored
    Wt = W.transpose(1, 2).contiguous()                                      # [K, 4H]
    G = torch.empty(cnt, rows, 4 * H, device=dev, dtype=torch.bfloat16)
    dG = torch.randn(cnt, rows, 4 * H, device=dev, dtype=torch.bfloat16)
    D = torch.empty(cnt, rows, H, device=dev, dtype=torch.bfloat16)
    b = torch.randn(cnt, 1, 4 * H, device=dev, dtype=torch.bfloat16)
    F = 2.0 * cnt * rows * H * 4 * H
    bench(f"fwd  {cnt}x[{rows}x{H}] . W^T   (W [4H,K]: NT)", lambda: torch.baddbmm(b, X, W.transpose(1, 2), out=G), F)
    bench(f"fwd  {cnt}x[{rows}x{H}] . Wt    (Wt [K,4H]: NN)", lambda: torch.baddbmm(b, X, Wt, out=G), F)
    bench(f"bwd  {cnt}x[{rows}x{4*H}] . W     (NN)", lambda: torch.bmm(dG, W, out=D), F)
    bench(f"bwd  {cnt}x[{rows}x{4*H}] . Wt^T  (NT)", lambda: torch.bmm(dG, Wt.transpose(1, 2), out=D), F)
