"""joint_fc weight gradient (304 000 x 8704 x 768) per slice count: the plan's choice against forced CAIMAN_WGRAD_SLICES values
(one child process per value: the library reads the variable once)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from caiman_asr_amd.train_utils.overlap import _joint_wgrad
M, K, N = 304000, 768, 8704
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
for _ in range(2): _joint_wgrad(dy, a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record(); _joint_wgrad(dy, a); _joint_wgrad(dy, a); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) / 2)
print(min(ts))
''' % ROOT
out = {}
for s in ("plan", "2", "3", "4", "5", "6", "8", "10", "15"):
    env = dict(os.environ)
    if s != "plan":
        env["CAIMAN_WGRAD_SLICES"] = s
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    out[s] = float(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else r.stderr[-200:]
print(json.dumps(out))
