#!/bin/bash
# ab_multi.sh <out> <ENVVAR> <v1> <v2> ... -- [bench args]: python bench.py under several values of one environment variable,
# two rounds, alternating, on ONE box (30 timed steps each).
set -o pipefail
O=gpurun_out/$1; VAR=$2; shift 2; VALS=()
while [ "$1" != "--" ] && [ -n "$1" ]; do VALS+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p $O
for r in 1 2; do
  for v in "${VALS[@]}"; do
    env $VAR=$v timeout -k 10 400 python bench.py --steps 30 --no-cpu-baseline --no-decode "$@" > $O/${VAR}_${v}_$r.json 2> $O/${VAR}_${v}_$r.err || { echo "$VAR=$v failed"; tail -5 $O/${VAR}_${v}_$r.err; exit 1; }
    python -c "
import json
d=json.loads(open('$O/${VAR}_${v}_$r.json').read().strip().splitlines()[-1])
print('$VAR=$v round $r:', round(d['ms_per_step'],3), 'ms', 'max step', max(d['step_ms_device']))"
  done
done
