#!/bin/bash
# r4_jgemm.sh <out>: joint projection GEMM parity tests, then (only if they pass: no further GPU step behind a failed one) the
# interleaved bench; KSWEEP=1 adds the fixed-cost / per-K-tile sweep
set -o pipefail
O=gpurun_out/$1; mkdir -p $O
[ -x tools/permlane_probe.bin ] && tools/permlane_probe.bin > $O/permlane.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_joint_gemm.py -x -q -m gpu > $O/test.log 2>&1 || { echo "tests failed"; tail -5 $O/test.log; exit 1; }
tail -2 $O/test.log
timeout -k 10 300 python tools/joint_gemm_bench.py --rounds 5 > $O/bench8.json 2> $O/bench8.err || { echo "bench failed"; tail -3 $O/bench8.err; exit 1; }
cat $O/bench8.json
if [ -n "$KSWEEP" ]; then
  timeout -k 10 300 python tools/joint_gemm_ksweep.py > $O/ksweep.json 2> $O/ksweep.err || { echo "ksweep failed"; exit 1; }
  cat $O/ksweep.json
fi
