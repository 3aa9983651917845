#!/bin/bash
# post layers' dR + dW as one launch of the weight-gradient kernel: tests, A/B in the step
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3ap
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_lstm.py tests/test_gpu_train_step.py tests/test_gpu_fullsize.py tests/test_gpu_model.py -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for v in 1 0 1 0 1 0; do
  CAIMAN_LSTM_WGRAD_BOTH=$v timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode --no-kernel-timing > $O/base_both${v}_$RANDOM.json 2> $O/base.err; echo "both=$v rc=$?"
done
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3ap/*.json")):
    d = json.load(open(f)); print(f, round(d["ms_per_step"], 2))
PY
