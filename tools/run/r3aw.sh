#!/bin/bash
# pipeline chunk by hidden size (24 / 32): tests, bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3aw
mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests/test_gpu_lstm.py tests/test_gpu_train_step.py tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_distributed.py tests/test_gpu_resident_oracle.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-200
for i in 1 2; do
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/base_$i.json 2> $O/base.err; echo "rc=$?"
done
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3aw/*.json")):
    d = json.load(open(f)); print(f, round(d["ms_per_step"], 2), round(d["value"], 3), d["lstm_resident"])
PY
