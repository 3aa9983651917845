#!/bin/bash
# ab.sh <out> <ENVVAR> <value A> <value B> [bench args...]: python bench.py under ENVVAR=A and ENVVAR=B, alternating, two rounds
# each on ONE box (ms_per_step printed per run).  Stops at the first failing run.
set -o pipefail
O=gpurun_out/$1; VAR=$2; A=$3; B=$4; shift 4; mkdir -p $O
for r in 1 2; do
  for v in "$A" "$B"; do
    env $VAR=$v timeout -k 10 400 python bench.py --no-cpu-baseline --no-decode "$@" > $O/${VAR}_${v}_$r.json 2> $O/${VAR}_${v}_$r.err || { echo "$VAR=$v failed"; tail -5 $O/${VAR}_${v}_$r.err; exit 1; }
    python -c "
import json,sys
d=json.loads(open('$O/${VAR}_${v}_$r.json').read().strip().splitlines()[-1])
print('$VAR=$v round $r:', round(d['ms_per_step'],3), 'ms', round(d['value'],4))"
  done
done
