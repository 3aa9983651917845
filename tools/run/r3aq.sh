#!/bin/bash
# fused SpecAugment + splice + permute kernel: tests, then the step
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3aq
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_specaugment.py tests/test_gpu_data_feed.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode --no-kernel-timing > $O/base_$i.json 2> $O/base.err; echo "rc=$?"
done
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3aq/*.json")):
    d = json.load(open(f)); print(f, round(d["ms_per_step"], 2), round(d["value"], 3), d["config"]["final_loss"])
PY
