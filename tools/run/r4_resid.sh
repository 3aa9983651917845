#!/bin/bash
# r4_resid.sh <out>: bf16 residual table with the HIP column, smoke(), and the oracle-anchored model tests
set -o pipefail
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 600 python tools/bf16_residual_table.py --hip > $O/resid_mfma.md 2> $O/resid.err || { echo "table failed"; tail -5 $O/resid.err; exit 1; }
tail -3 $O/resid_mfma.md
timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo "smoke failed"; tail -5 $O/smoke.log; exit 1; }
cat $O/smoke.log | grep "smoke ok"
timeout -k 10 900 python -m pytest tests/test_gpu_train_step.py -x -q -m gpu -k "config0" > $O/t.log 2>&1 || { echo "tests failed"; tail -8 $O/t.log; exit 1; }
tail -2 $O/t.log
