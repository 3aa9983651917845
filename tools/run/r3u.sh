#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3u
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_joint_gemm.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/joint_gemm_bench.py --rounds 3 --large > $O/jb_large.json 2> $O/jb.err; echo "jb large rc=$?"
python3 -c "
import json; d=json.load(open('$O/jb_large.json')); print({k: d[k] for k in ('hand_dw','lib_dw','max_abs_diff_dw_rel')})"
for w in 1 0; do
CAIMAN_JOINT_WGRAD=$w timeout -k 10 300 python3 bench.py --model large --steps 5 --warmup 2 --no-cpu-baseline --no-decode > $O/large_w$w.json 2> $O/large.err; echo "large w=$w rc=$?"
done
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3u/large*.json")):
    d = json.load(open(f)); print(f, round(d["ms_per_step"], 2), d["value"])
PY
