#!/bin/bash
# r4_jgemm.sh <out>: joint projection GEMM parity tests, then the interleaved bench for the 8-phase kernel and the ring kernel
set -o pipefail
O=gpurun_out/$1; mkdir -p $O
[ -x tools/permlane_probe.bin ] && tools/permlane_probe.bin > $O/permlane.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_joint_gemm.py -x -q -m gpu > $O/test.log 2>&1; echo "tests rc $?"; tail -5 $O/test.log
timeout -k 10 300 python tools/joint_gemm_bench.py --rounds 5 > $O/bench8.json 2> $O/bench8.err && cat $O/bench8.json
[ -n "$RING" ] && CAIMAN_JOINT_KERNEL=ring timeout -k 10 300 python tools/joint_gemm_bench.py --rounds 3 > $O/bench_ring.json 2> $O/bench_ring.err && cat $O/bench_ring.json
[ -n "$KSWEEP" ] && timeout -k 10 300 python tools/joint_gemm_ksweep.py > $O/ksweep.json 2> $O/ksweep.err && cat $O/ksweep.json
