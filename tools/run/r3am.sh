#!/bin/bash
# projection GEMM tile variants in the step: default (5 forward / 8 backward) vs the four-stage ring (10)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3am
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_proj_gemm.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for v in "0 8" "10 10" "0 10" "0 8" "10 10" "0 10"; do
  set -- $v
  CAIMAN_PROJ_TILE=$1 CAIMAN_PROJ_TILE_BWD=$2 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode --no-kernel-timing > $O/base_f$1_b$2_$RANDOM.json 2> $O/base.err; echo "$v rc=$?"
done
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3am/*.json")):
    d = json.load(open(f)); print(f, round(d["ms_per_step"], 2))
PY
