#!/bin/bash
# joint GEMM variants + tests + profile passes (base, B = 128)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_joint_gemm.py -x -q > $O/t_joint.log 2>&1; echo "joint tests rc=$?"; tail -3 $O/t_joint.log
for g in 1 8 16; do for p in 0 1; do
  CAIMAN_JOINT_GROUP=$g CAIMAN_JOINT_PRIO=$p timeout -k 10 300 python3 tools/joint_gemm_bench.py --rounds 3 > $O/jb_g${g}_p${p}.json 2> $O/jb.err; echo "g=$g p=$p: $(python3 -c "
import json; d=json.load(open('$O/jb_g${g}_p${p}.json')); print(d['hand_fwd_lse']['ms_median'], d['hand_dx']['ms_median'], d['lib_fwd_lse']['ms_median'], d['lib_dx']['ms_median'])")"
done; done
timeout -k 10 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_train_step.py tests/test_gpu_fullsize.py -x -q -k "oracle or storage" > $O/t_oracle.log 2>&1; echo "oracle tests rc=$?"; tail -4 $O/t_oracle.log
bash tools/run/profile_config.sh r3e/prof_base python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-decode; echo "prof base rc=$?"
bash tools/run/profile_config.sh r3e/prof_b128 python3 $R/bench.py --batch 128 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-decode; echo "prof b128 rc=$?"
du -sh $O
