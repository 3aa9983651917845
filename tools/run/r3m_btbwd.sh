#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3m
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_resident_oracle.py tests/test_gpu_fullsize.py -x -q > $O/t_res.log 2>&1; echo "resident+fullsize tests rc=$?"; tail -4 $O/t_res.log
for m in 1 0 1 0; do
  CAIMAN_BT_DMA=$m timeout -k 10 300 python3 - > $O/b128_dma${m}_$RANDOM.json 2> $O/b128_dma$m.err <<PY
import os, sys, runpy
from caiman_asr_amd import _lib
_lib.lib().caiman_lstm_resident_bt_dma(int(os.environ["CAIMAN_BT_DMA"]))
sys.argv = ["bench.py", "--batch", "128", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-decode"]
runpy.run_path("bench.py", run_name="__main__")
PY
  echo "b128 dma=$m rc=$?"
done
timeout -k 10 300 python3 bench.py --batch 64 --steps 6 --warmup 2 --no-cpu-baseline --no-decode > $O/b64.json 2> $O/b64.err; echo "b64 rc=$?"
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3m/b*.json")):
    try:
        d = json.load(open(f))
        print(f, round(d["ms_per_step"], 2), round(d["value"], 3), d.get("kernel_ms_per_step", {}).get("lstm_fwd"), d.get("kernel_ms_per_step", {}).get("lstm_bwd"), d["lstm_resident"])
    except Exception as e:
        print(f, "ERR", e)
PY
