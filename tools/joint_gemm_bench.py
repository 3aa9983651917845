"""The joint projection (rows x 768 x 8704, bf16) on the hand-written GEMM with the log-sum-exp epilogue
(csrc/joint_gemm.hip) against what it replaces: F.linear (hipBLASLt) + the row log-sum-exp kernel; and the input-gradient
product dY . W (rows x 8704 x 768) on the same kernel against torch.mm; and the weight gradient dY^T . h on
csrc/joint_wgrad.hip against the library's batched call.  Interleaved rounds in one process.
python tools/joint_gemm_bench.py [--rows 304000] [--rounds 5] [--large]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from caiman_asr_amd.rnnt_ext.cuda.logsumexp import logsumexp  # noqa: E402
from caiman_asr_amd.train_utils import overlap  # noqa: E402
from caiman_asr_amd.train_utils.overlap import _joint_gemm, _joint_wgrad, _weight_gradient  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=304000)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--large", action="store_true", help="large-196M shapes: K = 1024, N = 17408")
args = ap.parse_args()
M, K, N = args.rows, (1024 if args.large else 768), (17408 if args.large else 8704)
dev = "cuda"
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
b = torch.randn(N, device=dev).to(torch.bfloat16)
dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
wt = w.t().contiguous()


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


def lib_dw():
    """the library path of the weight gradient: 16 row chunks in one batched call (CAIMAN_JOINT_WGRAD=0)"""
    keep, overlap.JOINT_WGRAD = overlap.JOINT_WGRAD, False
    try:
        return _weight_gradient(dy, a)
    finally:
        overlap.JOINT_WGRAD = keep


def lib_fwd():
    c = torch.nn.functional.linear(a, w, b)
    return c, logsumexp(c, 128, True)


res = {"rows": M, "K": K, "N": N, "tflop": 2.0 * M * N * K / 1e12}
rows = {k: [] for k in ("hand_fwd_lse", "hand_fwd", "lib_fwd", "lib_fwd_lse", "hand_dx", "lib_dx", "hand_dw", "lib_dw")}
for _ in range(args.rounds):
    rows["hand_fwd_lse"].append(timed(lambda: _joint_gemm(a, w, b, True)))
    rows["hand_fwd"].append(timed(lambda: _joint_gemm(a, w, b, False)))
    rows["lib_fwd"].append(timed(lambda: torch.nn.functional.linear(a, w, b)))
    rows["lib_fwd_lse"].append(timed(lib_fwd))
    rows["hand_dx"].append(timed(lambda: _joint_gemm(dy, wt, None, False)))
    rows["lib_dx"].append(timed(lambda: torch.mm(dy, wt.t())))
    rows["hand_dw"].append(timed(lambda: _joint_wgrad(dy, a)))
    rows["lib_dw"].append(timed(lib_dw))
for k, v in rows.items():
    med = sorted(v)[len(v) // 2]
    res[k] = {"ms_median": round(med, 3), "ms_min": round(min(v), 3), "pflops_median": round(res["tflop"] / med, 3)}
c1, l1 = _joint_gemm(a, w, b, True)
c0, l0 = lib_fwd()
res["max_abs_diff_logits"] = float((c1.float() - c0.float()).abs().max())
res["max_abs_diff_lse"] = float((l1 - l0).abs().max())
dw1, dw0 = _joint_wgrad(dy, a), lib_dw().float()
res["max_abs_diff_dw_rel"] = float((dw1 - dw0).abs().max() / dw0.abs().max())
print(json.dumps(res))
