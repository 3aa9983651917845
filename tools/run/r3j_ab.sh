#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3j
mkdir -p $O
cd $R
for rep in 1 2; do for v in DEFAULT EPI_EARLY PART_SERIAL BOTH; do
  if [ $v = DEFAULT ]; then unset CAIMAN_LIB_OVERRIDE; else export CAIMAN_LIB_OVERRIDE=$R/caiman_asr_amd/lib/variants/libcaiman_$v.so; fi
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/base_${v}_$rep.json 2> $O/base.err; echo "$v rc=$?"
done; done
unset CAIMAN_LIB_OVERRIDE
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3j/*.json")):
    try:
        d = json.load(open(f))
        print(f, round(d["ms_per_step"], 2), d["kernel_ms_per_step"]["lstm_fwd"], d["kernel_ms_per_step"]["lstm_bwd"], d["lstm_resident"])
    except Exception as e:
        print(f, "ERR", e)
PY
