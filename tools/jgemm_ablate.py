"""Main-loop ablation of the joint projection kernel on its input-gradient instance (304 000 x 8704 -> 768, 136 K tiles per
output tile): the shipped library against measurement builds without the fragment reads / without the LDS-DMA issue
(tools/build_variant.py noreads|nodma joint_gemm.hip -DJG_NO_READS|-DJG_NO_DMA; their results are invalid by construction)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from caiman_asr_amd.train_utils.overlap import _joint_gemm
M, K, N = 304000, 8704, 768
dy = torch.randn(M, K, device="cuda").to(torch.bfloat16)
wt = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
for _ in range(2): _joint_gemm(dy, wt, None, False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record(); _joint_gemm(dy, wt, None, False); _joint_gemm(dy, wt, None, False); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) / 2)
print(min(ts))
''' % ROOT
out = {}
for name in ("shipped", "noreads", "nodma", "shipped"):
    env = dict(os.environ)
    if name != "shipped":
        env["CAIMAN_LIB_OVERRIDE"] = os.path.join(ROOT, "caiman_asr_amd", "lib", "variants", f"libcaiman_{name}.so")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    out.setdefault(name, []).append(float(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else r.stderr[-200:])
out["mfma_floor_ms_at_2.4GHz"] = 2 * 304000 * 8704 * 768 / 2.5e15 * 1e3
print(json.dumps(out))
