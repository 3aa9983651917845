"""joint_fc forward GEMM + log-sum-exp over its rows: back to back on one stream vs row chunks with the LSE of chunk i on a
second stream under the GEMM of chunk i + 1."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from caiman_asr_amd.rnnt_ext.cuda.logsumexp import logsumexp
dev = "cuda"
torch.manual_seed(0)
rows, V, H = 304128, 8704, 768
h = torch.relu(torch.randn(rows, H, device=dev)).to(torch.bfloat16)
Wt = (torch.randn(H, V, device=dev) * 0.03).to(torch.bfloat16)
b = torch.zeros(V, device=dev, dtype=torch.bfloat16)
logits = torch.empty(rows, V, device=dev, dtype=torch.bfloat16)
side = torch.cuda.Stream()
def timeit(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def serial():
    torch.addmm(b, h, Wt, out=logits)
    return logsumexp(logits, 128, True)
def chunked(nc):
    main = torch.cuda.current_stream()
    step = (rows + nc - 1) // nc
    outs = []
    for i in range(nc):
        r0, r1 = i * step, min(rows, (i + 1) * step)
        torch.addmm(b, h[r0:r1], Wt, out=logits[r0:r1])
        ev = torch.cuda.Event(); ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            outs.append(logsumexp(logits[r0:r1], 128, True))
    main.wait_stream(side)
    return outs
print("GEMM alone", timeit(lambda: torch.addmm(b, h, Wt, out=logits)))
print("LSE alone", timeit(lambda: logsumexp(logits, 128, True)))
print("serial", timeit(serial))
for nc in (2, 4, 8, 16):
    print("chunks", nc, timeit(lambda: chunked(nc)))
ref = serial(); got = torch.cat(chunked(4))
torch.cuda.synchronize()
print("equal", torch.equal(ref, got))
