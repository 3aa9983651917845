import sys, time, torch
rows, N = 300_000, 8704
for K in (768, 784, 800, 832):
    h = torch.randn(rows, K, device="cuda", dtype=torch.bfloat16)
    W = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    dY = torch.randn(rows, N, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(rows, N, device="cuda", dtype=torch.bfloat16)
    dX = torch.empty(rows, K, device="cuda", dtype=torch.bfloat16)
    dW = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
    def bench(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / 10 * 1e3
    print(K, "fwd %.3f  dX %.3f  dW %.3f ms" % (bench(lambda: torch.mm(h, W.t(), out=out)), bench(lambda: torch.mm(dY, W, out=dX)),
                                               bench(lambda: torch.mm(dY.t(), h, out=dW))))
    del h, W, dY, out, dX, dW
x = torch.randn(300_000, 8704, device="cuda", dtype=torch.bfloat16)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): x.sum(0)
torch.cuda.synchronize(); print("bias-grad reduce %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
