import time, torch
dev = "cuda"
x = torch.randn(1024, 1024, device=dev, dtype=torch.bfloat16)
W = torch.randn(4096, 1024, device=dev, dtype=torch.bfloat16)
b = torch.randn(4096, device=dev, dtype=torch.bfloat16)
G = torch.empty(8, 1024, 4096, device=dev, dtype=torch.bfloat16)
def bench(f, n=300):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
print("addmm out=      cpu/total us", bench(lambda: torch.addmm(b, x, W.t(), out=G[3])))
print("addmm           cpu/total us", bench(lambda: torch.addmm(b, x, W.t())))
print("matmul          cpu/total us", bench(lambda: torch.matmul(x, W.t())))
print("F.linear        cpu/total us", bench(lambda: torch.nn.functional.linear(x, W, b)))
y = torch.randn(32, 32, 1024, device=dev, dtype=torch.bfloat16); m = torch.randn_like(y)
print("mul             cpu/total us", bench(lambda: y * m))
with torch.autocast("cuda", dtype=torch.bfloat16):
    print("autocast addmm out= ", bench(lambda: torch.addmm(b, x, W.t(), out=G[3])))
    print("autocast addmm      ", bench(lambda: torch.addmm(b, x, W.t())))
