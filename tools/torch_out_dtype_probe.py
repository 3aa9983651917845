"""Host and device cost of the torch GEMM variants the direct-gradient paths use (run on the GPU box)."""
import time

import torch

d = "cuda"
M, N, K = 6400, 512, 1024
dy = torch.randn(M, N, device=d).bfloat16()
x = torch.randn(M, K, device=d).bfloat16()
g = torch.zeros(N, K, device=d)
gb = torch.zeros(N, 1, device=d)
ones = torch.ones(M, 1, device=d).bfloat16()
cases = {
    "mm out_dtype": lambda: torch.mm(dy.t(), x, out_dtype=torch.float32),
    "addmm out_dtype out=grad": lambda: torch.addmm(g, dy.t(), x, out_dtype=torch.float32, out=g),
    "addmm ones column": lambda: torch.addmm(gb, dy.t(), ones, out_dtype=torch.float32, out=gb),
    "sum(0, f32)": lambda: dy.sum(0, dtype=torch.float32),
}
for name, fn in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:28s} host {1e6 * (t1 - t0) / 50:8.1f} us/call, total {1e6 * (t2 - t0) / 50:8.1f} us/call")
