#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/base.json 2> $O/base.err; echo "base rc=$?"
timeout -k 10 300 python3 bench.py --model large --steps 8 --warmup 2 --no-decode > $O/large.json 2> $O/large.err; echo "large rc=$?"
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3g/*.json")):
    try:
        d = json.load(open(f))
        print(f, round(d["ms_per_step"], 2), round(d["value"], 3), d.get("kernel_ms_per_step"), d["lstm_resident"], d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/t_all.log 2>&1; echo "ALL gpu tests rc=$?"; tail -5 $O/t_all.log
