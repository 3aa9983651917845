#!/bin/bash
# r4_slices.sh <out>: the batch-slice launches of the B <= 32 resident LSTM kernels (csrc/lstm.hip::res_batch_slice):
# their tests first, then an A/B against the per-timestep kernels (CAIMAN_LSTM_BATCH_SLICES=0) on large-196M at 128 per GPU
# (the A/B of the round, 189.6 -> 184.2 ms, was against Python-level 32-utterance chunks that have since been removed).
set -o pipefail
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_resident_oracle.py tests/test_gpu_lstm.py -m gpu -x -q > $O/tests.log 2>&1 || { echo "tests failed"; tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
bash tools/run/ab.sh $1 CAIMAN_LSTM_BATCH_SLICES 0 1 --model large --batch 128 --steps 6 --warmup 2 || exit 1
