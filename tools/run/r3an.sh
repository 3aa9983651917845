#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3an
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_proj_gemm.py tests/test_gpu_train_step.py tests/test_gpu_fullsize.py tests/test_gpu_lstm.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-decode > $O/b128.json 2> $O/b128.err
CAIMAN_PROJ_TILE_BWD=8 timeout -k 10 300 python3 bench.py --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-decode > $O/b128_t8.json 2> $O/b128.err
timeout -k 10 300 python3 bench.py --model large --steps 5 --warmup 2 --no-cpu-baseline --no-decode > $O/large.json 2> $O/large.err
CAIMAN_PROJ_TILE_BWD=8 timeout -k 10 300 python3 bench.py --model large --steps 5 --warmup 2 --no-cpu-baseline --no-decode > $O/large_t8.json 2> $O/large.err
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3an/*.json")):
    d = json.load(open(f)); print(f, round(d["ms_per_step"], 2), round(d["value"], 3))
PY
