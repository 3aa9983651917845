#!/bin/bash
# ab_tree.sh <out> <other tree dir> [bench args...]: bench.py of this tree and of another checkout (with its own built library),
# alternating, two rounds each on ONE box
set -o pipefail
O=$PWD/gpurun_out/$1; OTHER=$2; shift 2; mkdir -p $O
HERE=$PWD
for r in 1 2; do
  for t in "$HERE" "$HERE/$OTHER"; do
    tag=$(basename $t)
    ( cd $t && timeout -k 10 400 python bench.py --no-cpu-baseline --no-decode "$@" > $O/${tag}_$r.json 2> $O/${tag}_$r.err ) || { echo "$tag failed"; tail -5 $O/${tag}_$r.err; exit 1; }
    python -c "
import json
d=json.loads(open('$O/${tag}_$r.json').read().strip().splitlines()[-1])
print('$tag round $r:', round(d['ms_per_step'],3), 'ms')"
  done
done
