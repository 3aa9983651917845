#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/run/profile_config.sh r3h/prof_b128 python3 $R/bench.py --batch 128 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-decode; echo "prof b128 rc=$?"
bash tools/run/profile_config.sh r3h/prof_large python3 $R/bench.py --model large --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-decode; echo "prof large rc=$?"
bash tools/run/profile_config.sh r3h/prof_decode python3 $R/bench_decode.py --decoder beam --streams 2000 --ticks 30 --warmup 10 --from-audio --scale 5562.699766687201 --blank-bias 466.94; echo "prof decode rc=$?"
du -sh $R/gpurun_out/r3h
