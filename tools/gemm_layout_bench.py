"""Which operand layout does the BLAS library like for the joint's three GEMMs?  rows x 768 x 8704, bf16.
python tools/gemm_layout_bench.py [rows]"""
import sys
import time

import torch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
K, N = 768, 8704
dev = "cuda"
h = torch.randn(rows, K, device=dev, dtype=torch.bfloat16)
W = torch.randn(N, K, device=dev, dtype=torch.bfloat16)      # nn.Linear layout
Wt = W.t().contiguous()                                        # [K, N]
dY = torch.randn(rows, N, device=dev, dtype=torch.bfloat16)
bias = torch.randn(N, device=dev, dtype=torch.bfloat16)
out = torch.empty(rows, N, device=dev, dtype=torch.bfloat16)
dX = torch.empty(rows, K, device=dev, dtype=torch.bfloat16)
dW = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
dWt = torch.empty(K, N, device=dev, dtype=torch.bfloat16)
ht = h.t().contiguous()
dYt = None


def bench(name, fn, flops):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{name:48s} {dt * 1e3:7.3f} ms  {flops / dt / 1e12:7.1f} TFLOP/s")


F = 2.0 * rows * K * N
bench("fwd  linear(h, W) + bias            (NT)", lambda: torch.addmm(bias, h, W.t(), out=out), F)
bench("fwd  h @ Wt + bias                  (NN)", lambda: torch.addmm(bias, h, Wt, out=out), F)
bench("fwd  no bias, mm(h, W.t())", lambda: torch.mm(h, W.t(), out=out), F)
bench("dX   dY @ W                         (NN)", lambda: torch.mm(dY, W, out=dX), F)
bench("dX   dY @ Wt.t()                    (NT)", lambda: torch.mm(dY, Wt.t(), out=dX), F)
bench("dW   dY.t() @ h                     (TN)", lambda: torch.mm(dY.t(), h, out=dW), F)
bench("dWt  h.t() @ dY                     (TN)", lambda: torch.mm(h.t(), dY, out=dWt), F)
bench("dWt  ht @ dY  (h pre-transposed)    (NN)", lambda: torch.mm(ht, dY, out=dWt), F)
f32 = torch.empty(N, K, device=dev, dtype=torch.float32)
bench("dW   fp32 out: (dY.t() @ h).float()", lambda: f32.copy_(torch.mm(dY.t(), h)), F)
