#!/bin/bash
# round-3 baseline of the round-2 tree on this round's box: bench lines + rocprofv3 kernel traces for B=128, large, decode tick
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3a
mkdir -p $O
export TMPDIR=/tmp
cd $R
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/base.json 2> $O/base.err && echo base done
python3 bench.py --batch 128 --steps 6 --warmup 2 --no-cpu-baseline --no-decode > $O/b128.json 2> $O/b128.err && echo b128 done
python3 bench.py --model large --steps 6 --warmup 2 --no-decode > $O/large.json 2> $O/large.err && echo large done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b128 -- python3 $R/bench.py --batch 128 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-decode > $O/prof_b128.log 2>&1 && echo prof b128 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_large -- python3 $R/bench.py --model large --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-decode > $O/prof_large.log 2>&1 && echo prof large done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_decode -- python3 $R/bench_decode.py --decoder beam --streams 2000 --ticks 40 --warmup 10 --from-audio --scale 5562.699766687201 --blank-bias 466.94 > $O/prof_decode.log 2>&1 && echo prof decode done
# keep only the stats csvs (the traces are large)
find $O -name '*kernel_trace.csv' -size +20M -delete
ls -la $O $O/prof_*/*/ 2>/dev/null | head -60
