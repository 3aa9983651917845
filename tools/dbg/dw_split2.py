import torch
torch.manual_seed(0)
dev = "cuda"
rows, V, H = 304128, 8704, 768
dy = torch.randn(rows, V, device=dev, dtype=torch.bfloat16) * 0.01
x = torch.randn(rows, H, device=dev, dtype=torch.bfloat16)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
flops = 2.0 * rows * V * H
def rep(name, ms): print(f"{name:44s} {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TF/s", flush=True)
for S in (8, 11, 12, 16, 22, 24, 32, 44, 48, 64):
    r = rows // S * S
    a = dy[:r].view(S, r // S, V); b = x[:r].view(S, r // S, H)
    rep(f"bmm split {S} (dy^T x) + sum", timeit(lambda: torch.bmm(a.transpose(1, 2), b).sum(0)))
    try:
        rep(f"bmm split {S} fp32 out + sum", timeit(lambda: torch.bmm(a.transpose(1, 2), b, out_dtype=torch.float32).sum(0)))
    except Exception as e:
        print("out_dtype unsupported:", repr(e)[:100])
ref = torch.mm(dy.t().float()[:, :20000], x.float()[:20000])
S = 8
print("done")
