"""Fixed cost vs per-K-tile cost of the joint projection kernel: rows x K x 8704 at several K (with / without the log-sum-exp
epilogue), and the library beside it.  time(K) = rounds x (fixed + K / 64 x per_tile)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from caiman_asr_amd.train_utils.overlap import _joint_gemm  # noqa: E402

M, N = 304000, 8704
dev = "cuda"


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


out = {}
for K in (128, 256, 512, 768, 1536, 3072):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    r = {}
    for _ in range(3):
        r.setdefault("hand", []).append(timed(lambda: _joint_gemm(a, w, b, False)))
        r.setdefault("hand_lse", []).append(timed(lambda: _joint_gemm(a, w, b, True)))
        r.setdefault("lib", []).append(timed(lambda: torch.nn.functional.linear(a, w, b)))
    out[K] = {k: round(min(v), 3) for k, v in r.items()}
rounds = ((M + 255) // 256) * (N // 256) / 256.0
out["rounds_of_256_tiles"] = rounds
for k in ("hand", "hand_lse", "lib"):
    slope = (out[3072][k] - out[768][k]) / ((3072 - 768) / 64)
    out[f"{k}_us_per_ktile_per_round"] = round(slope / rounds * 1e3, 3)
    out[f"{k}_fixed_us_per_round"] = round((out[768][k] - slope * 12) / rounds * 1e3, 3)
print(json.dumps(out))
