"""Where does one beam expansion round go?  Wall-clock per stage of HipBeamStep (synchronising between stages)
and end-to-end per round for several request counts.  python tools/beam_round_profile.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import BASE_RNNT, N_CLASSES  # noqa: E402
from caiman_asr_amd.rnnt.beam_native import HipBeamStep  # noqa: E402
from caiman_asr_amd.rnnt.model import RNNT  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = RNNT(n_classes=N_CLASSES, **dict(BASE_RNNT, joint_apex_transducer=None, joint_apex_relu_dropout=False)).to(dev).eval()
step = HipBeamStep(model, N_CLASSES - 1, 4, 1.4)
rng = np.random.default_rng(0)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for n in (2000, 512, 64, 4):
        frames = torch.randn(4096, 768, device=dev, dtype=torch.bfloat16)
        rows = rng.integers(0, 4096, n).astype(np.int64)
        y = rng.integers(1, 8000, n).astype(np.int32)
        s_in = rng.integers(0, 4000, n).astype(np.int32)
        s_out = (4000 + np.arange(n)).astype(np.int32)
        for _ in range(5):
            step(frames, rows, y, s_in, s_out, 8000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            step(frames, rows, y, s_in, s_out, 8000)
        torch.cuda.synchronize()
        print(f"n={n}: {1e3 * (time.perf_counter() - t0) / 50:.3f} ms per round")
