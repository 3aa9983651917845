#!/bin/bash
# one dropout hash per four elements: tests, then A/B against the previous library (variants/libcaiman_OLDRNG.so)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3z
mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q -k "not fullsize" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit $rc
for v in NEW OLDRNG NEW OLDRNG NEW OLDRNG; do
  if [ $v = NEW ]; then unset CAIMAN_LIB_OVERRIDE; else export CAIMAN_LIB_OVERRIDE=$R/caiman_asr_amd/lib/variants/libcaiman_$v.so; fi
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/base_${v}_$RANDOM.json 2> $O/base.err; echo "$v rc=$?"
done
unset CAIMAN_LIB_OVERRIDE
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3z/*.json")):
    try:
        d = json.load(open(f)); k = d["kernel_ms_per_step"]; print(f, round(d["ms_per_step"], 2), round(d["value"], 3), k["lstm_fwd"], k["lstm_bwd"], k["joint_fwd"])
    except Exception as e:
        print(f, "ERR", e)
PY
