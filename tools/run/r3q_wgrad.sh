#!/bin/bash
# weight-gradient kernel (csrc/joint_wgrad.hip): tests, then timing against the library's dW
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3q
mkdir -p $O
cd $R
timeout -k 10 400 python3 -m pytest tests/test_gpu_joint_gemm.py -x -q -k wgrad > $O/t_wgrad.log 2>&1; rc=$?; echo "wgrad tests rc=$rc"; tail -15 $O/t_wgrad.log
[ $rc -eq 0 ] || exit $rc
for s in ${SLICES:-0}; do
  CAIMAN_WGRAD_SLICES=$s timeout -k 10 300 python3 tools/joint_gemm_bench.py --rounds 3 > $O/jb_s$s.json 2> $O/jb_s$s.err || { echo "bench s=$s failed"; tail -5 $O/jb_s$s.err; exit 1; }
  echo "slices=$s: $(python3 -c "
import json; d=json.load(open('$O/jb_s$s.json')); print(d['hand_dw'], d['lib_dw'], d['max_abs_diff_dw_rel'])")"
done
