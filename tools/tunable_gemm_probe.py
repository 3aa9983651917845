"""Does PyTorch's TunableOp (a timed search over the hipBLASLt / rocBLAS solutions of one GEMM shape) find faster library
kernels than the default heuristic for the joint projection's products?  Times each product before tuning, tunes it, times it
again, and leaves the selection file.  python tools/tunable_gemm_probe.py [--rows 304000] [--out gpurun_out/tunable.csv]"""
import argparse
import json
import os
import sys
import time

import torch
import torch.cuda.tunable as tun

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=304000)
ap.add_argument("--out", default="gpurun_out/tunable.csv")
ap.add_argument("--large", action="store_true")
args = ap.parse_args()
M, K, N = args.rows, (1024 if args.large else 768), (17408 if args.large else 8704)
dev = "cuda"
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
b = torch.randn(N, device=dev).to(torch.bfloat16)
dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
wt = w.t().contiguous()
ops = {
    "fwd_linear_bias": lambda: torch.nn.functional.linear(a, w, b),
    "fwd_linear": lambda: torch.nn.functional.linear(a, w),
    "dx_nt": lambda: torch.mm(dy, wt.t()),
    "dx_nn": lambda: torch.mm(dy, w),
}


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


res = {"rows": M, "K": K, "N": N, "before_ms": {k: round(timed(f), 3) for k, f in ops.items()}}
print(json.dumps(res), flush=True)
os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
tun.enable(True)
tun.tuning_enable(True)
tun.set_filename(args.out)
tun.set_max_tuning_duration(40)
tun.set_max_tuning_iterations(10)
res["tune_s"] = {}
for k, f in ops.items():
    t0 = time.time()
    f()
    torch.cuda.synchronize()
    res["tune_s"][k] = round(time.time() - t0, 1)
    print(k, "tuned in", res["tune_s"][k], "s", flush=True)
tun.tuning_enable(False)
res["after_ms"] = {k: round(timed(f), 3) for k, f in ops.items()}
res["results"] = [list(map(str, r)) for r in tun.get_results()]
print(json.dumps(res))
