#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3l
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_decode.py -x -q > $O/t_decode.log 2>&1; echo "decode tests rc=$?"; tail -3 $O/t_decode.log
for f in 1 0 1 0; do
  CAIMAN_BEAM_SPIN=$f timeout -k 10 300 python3 bench_decode.py --decoder beam --streams 2000 --ticks 80 --warmup 10 --from-audio --scale 5562.699766687201 --blank-bias 466.94 --profile-host > $O/beam_spin${f}_$RANDOM.log 2>&1; echo "beam spin=$f rc=$?"
done
grep -h "host profile" $O/beam_spin*.log | cut -c1-300
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3l/beam*.log")):
    for ln in open(f):
        if ln.startswith("{"):
            d = json.loads(ln); print(f, d["tick_latency_ms"])
PY
