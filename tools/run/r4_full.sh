#!/bin/bash
# r4_full.sh <out> [bench args...]: the whole -m gpu suite, then (only if green) the default bench
set -o pipefail
O=gpurun_out/$1; shift; mkdir -p $O
timeout -k 10 1050 python -m pytest tests -x -q -m gpu --durations=25 > $O/gputests.log 2>&1 || { echo "gpu tests failed"; tail -15 $O/gputests.log; exit 1; }
tail -32 $O/gputests.log
timeout -k 10 400 python bench.py "$@" > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "roofline") if k in d})
PY
