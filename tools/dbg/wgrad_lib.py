"""joint_fc weight gradient (16 row chunks, fp32 partial outputs) and forward / input-gradient GEMMs under both BLAS back ends."""
import torch
torch.manual_seed(0)
dev = "cuda"
rows, V, H, S = 304128, 8704, 768, 16
dy = torch.randn(rows, V, device=dev, dtype=torch.bfloat16) * 0.01
x = torch.randn(rows, H, device=dev, dtype=torch.bfloat16)
W = torch.randn(V, H, device=dev, dtype=torch.bfloat16) * 0.03
Wt = W.t().contiguous()
b = torch.zeros(V, device=dev, dtype=torch.bfloat16)
def timeit(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
flops = 2.0 * rows * V * H
r = rows // S * S
a3, b3 = dy[:r].view(S, r // S, V), x[:r].view(S, r // S, H)
for lib in ("cublaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    for name, fn in (("wgrad bmm fp32 out + sum", lambda: torch.bmm(a3.transpose(1, 2), b3, out_dtype=torch.float32).sum(0)),
                     ("wgrad bmm bf16 out + fp32 sum", lambda: torch.bmm(a3.transpose(1, 2), b3).float().sum(0)),
                     ("wgrad mm", lambda: torch.mm(dy.t(), x)),
                     ("forward addmm", lambda: torch.addmm(b, x, Wt)),
                     ("dgrad mm", lambda: torch.mm(dy, W))):
        try:
            ms = timeit(fn)
            print(f"{lib:9s} {name:32s} {ms:7.3f} ms {flops / ms / 1e9:7.1f} TF/s", flush=True)
        except Exception as e:
            print(lib, name, "failed:", repr(e)[:120], flush=True)
