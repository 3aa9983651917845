import torch, time
torch.manual_seed(0)
dev = "cuda"
rows, V, H = 304128, 8704, 768
dy = torch.randn(rows, V, device=dev, dtype=torch.bfloat16) * 0.01
x = torch.randn(rows, H, device=dev, dtype=torch.bfloat16)
w = torch.randn(V, H, device=dev, dtype=torch.bfloat16) * 0.02

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

flops = 2.0 * rows * V * H
def rep(name, ms): print(f"{name:40s} {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TF/s", flush=True)

rep("dW = dy.t() @ x", timeit(lambda: torch.mm(dy.t(), x)))
rep("dW' = x.t() @ dy  ([H,V])", timeit(lambda: torch.mm(x.t(), dy)))
for S in (2, 3, 4, 6, 8):
    r = rows // S * S
    a = dy[:r].view(S, r // S, V)
    b = x[:r].view(S, r // S, H)
    rep(f"bmm split {S} (dy^T x) + sum", timeit(lambda: torch.bmm(a.transpose(1, 2), b).sum(0)))
    rep(f"bmm split {S} (x^T dy) + sum", timeit(lambda: torch.bmm(b.transpose(1, 2), a).sum(0)))
    out = torch.empty(S, V, H, device=dev, dtype=torch.float32)
rep("fwd  x @ w.t()", timeit(lambda: torch.mm(x, w.t())))
wt = w.t().contiguous()
rep("dX  dy @ w", timeit(lambda: torch.mm(dy, w)))
rep("dX  dy @ wt.t()", timeit(lambda: torch.mm(dy, wt.t())))
# fp32 output variants
rep("dW fp32 out via out_dtype? (bf16 default)", timeit(lambda: torch.mm(dy.t(), x)))
