#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/run/profile_config.sh r3o/prof_base python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-decode; echo "prof base rc=$?"
bash tools/run/profile_config.sh r3o/prof_b128 python3 $R/bench.py --batch 128 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-decode; echo "prof b128 rc=$?"
bash tools/run/profile_config.sh r3o/prof_large python3 $R/bench.py --model large --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-decode; echo "prof large rc=$?"
du -sh $R/gpurun_out/r3o
