#!/bin/bash
# bt_dma forward kernel: gather pieces interleaved with the MFMA blocks (default) vs all in front (variant UPFRONT)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3ai
mkdir -p $O
cd $R
if [ -z "$SKIP_TESTS" ]; then timeout -k 10 900 python3 -m pytest tests/test_gpu_resident_oracle.py -x -q -k "128 or bt or tile or 64 or 100 or 33" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/tests.log; else rc=0; fi
[ $rc -eq 0 ] || exit $rc
for v in ${VARIANTS:-NEW UPFRONT NEW UPFRONT}; do
  if [ $v = NEW ]; then unset CAIMAN_LIB_OVERRIDE; else export CAIMAN_LIB_OVERRIDE=$R/caiman_asr_amd/lib/variants/libcaiman_$v.so; fi
  timeout -k 10 300 python3 tools/lstm_resident_bench.py --skip-agreement --batch 128 > $O/rb_${v}_$RANDOM.log 2>&1; echo "$v rc=$?"
done
export CAIMAN_LIB_OVERRIDE=$R/caiman_asr_amd/lib/variants/libcaiman_BTPROF.so
timeout -k 10 300 python3 tools/lstm_resident_bench.py --skip-agreement --batch 128 > $O/rb_BTPROF.log 2>&1; echo "BTPROF rc=$?"
unset CAIMAN_LIB_OVERRIDE
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3ai/rb_*.log")):
    for l in open(f):
        if l.startswith('{"step_ms"'):
            d = json.loads(l); print(f, "resident_ms", round(d["resident_ms"], 3), "fwd_bwd", round(d["resident_fwd_bwd_ms"], 3), d.get("fwd_us_per_timestep"))
PY
