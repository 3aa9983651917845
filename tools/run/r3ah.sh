#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3ah
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_train_step.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/base_$i.json 2> $O/base.err; echo "rc=$?"
done
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3ah/*.json")):
    d = json.load(open(f)); k = d["kernel_ms_per_step"]; print(f, round(d["ms_per_step"], 2), round(d["value"], 3), k)
PY
