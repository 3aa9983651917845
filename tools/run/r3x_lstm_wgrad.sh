#!/bin/bash
# LSTM weight gradients on the transposed-read kernel: tests, then A/B in the step
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3x
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_joint_gemm.py tests/test_gpu_train_step.py tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_distributed.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
for w in 1 0 1 0 1 0; do
  CAIMAN_LSTM_WGRAD_TN=$w timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/base_w${w}_$RANDOM.json 2> $O/base.err; echo "w=$w rc=$?"
done
for w in 1 0; do
CAIMAN_LSTM_WGRAD_TN=$w timeout -k 10 300 python3 bench.py --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --no-decode > $O/b128_w$w.json 2> $O/b128.err; echo "b128 w=$w rc=$?"
CAIMAN_LSTM_WGRAD_TN=$w timeout -k 10 300 python3 bench.py --model large --steps 5 --warmup 2 --no-cpu-baseline --no-decode > $O/large_w$w.json 2> $O/large.err; echo "large w=$w rc=$?"
done
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3x/*.json")):
    try:
        d = json.load(open(f)); print(f, round(d["ms_per_step"], 2), round(d["value"], 3))
    except Exception as e:
        print(f, "ERR", e)
PY
