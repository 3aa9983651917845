"""A few launches of the joint projection's weight-gradient kernel alone (csrc/joint_wgrad.hip) at training shapes, for
rocprofv3 passes.  python tools/wgrad_once.py [--rows 304000] [--launches 3] [--large]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from caiman_asr_amd.train_utils.overlap import _joint_wgrad  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=304000)
ap.add_argument("--launches", type=int, default=3)
ap.add_argument("--large", action="store_true")
args = ap.parse_args()
M, K, N = args.rows, (1024 if args.large else 768), (17408 if args.large else 8704)
h = torch.randn(M, K, device="cuda").to(torch.bfloat16)
dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
for _ in range(args.launches):
    dw = _joint_wgrad(dy, h)
torch.cuda.synchronize()
print(float(dw.abs().max()))
