#!/bin/bash
# r4_sweep.sh <out> [test files...]: the given GPU test files (default: the ones the launch sweep touches), then one bench run.
set -o pipefail
O=gpurun_out/$1; shift; mkdir -p $O
T=${@:-tests/test_gpu_joint_gemm.py tests/test_gpu_specaugment.py tests/test_gpu_model.py tests/test_gpu_lstm.py tests/test_gpu_train_step.py}
timeout -k 10 1000 python -m pytest $T -m gpu -x -q > $O/tests.log 2>&1 || { echo "tests failed"; tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
bash tools/run/r4_bench.sh $(basename $O) --no-cpu-baseline --no-decode || exit 1
