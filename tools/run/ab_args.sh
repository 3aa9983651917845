#!/bin/bash
# ab_args.sh <out> "<args A>" "<args B>" [common bench args...]: python bench.py with argument set A and B, alternating, two rounds
# each on ONE box (ms_per_step printed per run).  Stops at the first failing run.
set -o pipefail
O=gpurun_out/$1; A=$2; B=$3; shift 3; mkdir -p $O
for r in 1 2; do
  i=0
  for v in "$A" "$B"; do
    i=$((i+1))
    timeout -k 10 400 python bench.py --no-cpu-baseline --no-decode $v "$@" > $O/args${i}_$r.json 2> $O/args${i}_$r.err || { echo "[$v] failed"; tail -5 $O/args${i}_$r.err; exit 1; }
    python -c "
import json,sys
d=json.loads(open('$O/args${i}_$r.json').read().strip().splitlines()[-1])
print('[$v] round $r:', round(d['ms_per_step'],3), 'ms', round(d['value'],4), d.get('host_ms_per_step'))"
  done
done
