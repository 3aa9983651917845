#!/bin/bash
# r4_bench.sh <out> [bench args...]: one bench.py run, JSON line kept under gpurun_out/<out>/
set -o pipefail
O=gpurun_out/$1; shift; mkdir -p $O
N=$(ls $O | wc -l)
timeout -k 10 500 python bench.py "$@" > $O/bench_$N.json 2> $O/bench_$N.err || { echo "bench failed"; tail -8 $O/bench_$N.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$O/bench_$N.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "roofline_step", "host_ms_per_step") if k in d})
print(d["config"]["timed_step"])
PY
