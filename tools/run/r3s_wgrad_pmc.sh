#!/bin/bash
# counters of the weight-gradient kernel alone: HBM-side fetch, L2 hit / miss, LDS conflicts
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3s
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/wgrad_once.py > $O/stats.log 2>&1 || { echo "stats failed"; tail -5 $O/stats.log; exit 1; }
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum"; do
  d=$(echo $c | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$d -- python3 $R/tools/wgrad_once.py > $O/$d.log 2>&1 || { echo "$c pass failed"; tail -5 $O/$d.log; continue; }
  find $O/$d -name '*kernel_trace.csv' -delete
  echo "$c done"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "joint_wgrad" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, [round(x) for x in v])
for f in glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "joint_wgrad" in r["Name"]:
            print("avg ns", r["AverageNs"], "calls", r["Calls"])
PY
