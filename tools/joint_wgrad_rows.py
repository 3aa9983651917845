"""The joint projection's weight gradient (csrc/joint_wgrad.hip through train_utils/overlap.py::_joint_wgrad) at a list of row
counts: ms per call, with the slice plan the library chose.  python tools/joint_wgrad_rows.py [--large] rows [rows ...]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from caiman_asr_amd import _lib  # noqa: E402
from caiman_asr_amd.train_utils.overlap import _joint_wgrad  # noqa: E402

large = "--large" in sys.argv
K, N = (1024, 17408) if large else (768, 8704)
for M in [int(a) for a in sys.argv[1:] if not a.startswith("--")]:
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    per = ctypes.c_int64(0)
    s = _lib.lib().caiman_wgrad_tn_plan(M, N, K, 1, _lib.dtype_tag(torch.bfloat16), ctypes.byref(per))
    ts = []
    for _ in range(3):
        _joint_wgrad(dy, a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            _joint_wgrad(dy, a)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 3)
    print(f"rows {M}: {s} slices of {per.value} rows, {sorted(ts)[1]:.3f} ms (min {min(ts):.3f})", flush=True)
    del dy, a
