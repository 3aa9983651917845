"""The LSTM layers' weight-gradient products on the transposed-read kernel (caiman_wgrad_tn) against the library's batched
transposed-A GEMM, at the shapes of the base encoder.  python tools/wgrad_tn_bench.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from caiman_asr_amd.train_utils.overlap import wgrad_tn  # noqa: E402


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res = []
for P, M, N, K in [(6, 8896, 4096, 1024), (5, 8896, 4096, 1024), (1, 17792, 4096, 1024), (1, 8896, 4096, 2048),
                   (6, 35584, 4096, 1024), (1, 71168, 4096, 1024), (2, 1920, 3072, 768)]:
    dy = torch.randn(P, M, N, device="cuda").to(torch.bfloat16)
    x = torch.randn(P, M, K, device="cuda").to(torch.bfloat16)
    t_hand = timed(lambda: wgrad_tn(dy, x))
    t_lib = timed(lambda: torch.bmm(dy.transpose(1, 2), x))
    t_lib32 = timed(lambda: torch.bmm(dy.transpose(1, 2), x, out_dtype=torch.float32))
    fl = 2.0 * P * M * N * K
    res.append({"P": P, "M": M, "N": N, "K": K, "hand_us": round(t_hand, 1), "lib_bf16_us": round(t_lib, 1), "lib_f32_us": round(t_lib32, 1),
                "hand_pf": round(fl / t_hand / 1e9, 3), "lib_pf": round(fl / t_lib / 1e9, 3)})
    print(json.dumps(res[-1]), flush=True)
