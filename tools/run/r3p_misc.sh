#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3p
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_resident_oracle.py -x -q -k "busy_side_stream or bit_identical" > $O/t_contend.log 2>&1; echo "contended test rc=$?"; tail -3 $O/t_contend.log
timeout -k 10 300 python3 -m pytest tests/test_gpu_joint_gemm.py -x -q > $O/t_joint.log 2>&1; echo "joint tests rc=$?"; tail -2 $O/t_joint.log
export CAIMAN_JOINT_GEMM=1
bash tools/run/profile_config.sh r3p/prof_jointgemm python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-decode; echo "prof jointgemm rc=$?"
unset CAIMAN_JOINT_GEMM
bash tools/run/profile_config.sh r3p/prof_decode python3 $R/bench_decode.py --decoder beam --streams 2000 --ticks 30 --warmup 10 --from-audio --scale 5562.699766687201 --blank-bias 466.94; echo "prof decode rc=$?"
du -sh $O
