#!/bin/bash
# profile_config.sh <out dir name under gpurun_out> <program> <args...>: kernel-trace pass + four one-counter PMC passes of
# the SAME command (same launches in every pass: tools/profile_summary.py checks it).  rocprofv3 wants the program itself
# after `--` (python3 ...), and the counter passes must not be combined with hip/hsa/memory traces.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1
shift
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- "$@" > $O/stats.log 2>&1 || { echo "stats pass failed"; tail -5 $O/stats.log; exit 1; }
find $O/stats -name '*kernel_trace.csv' -size +30M -delete
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -- "$@" > $O/$c.log 2>&1 || { echo "$c pass failed"; tail -5 $O/$c.log; exit 1; }
  find $O/$c -name '*kernel_trace.csv' -delete
  echo "$c done"
done
