#!/bin/bash
# r4_profiles.sh <which>: the rocprofv3 passes of one configuration (kernel trace + four one-counter PMC passes, identical
# arguments): base | b128 | large | large128 | decode
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
# warm-up 3 + 6 timed steps = one pass over the six synthetic batches behind the warm-up, i.e. the batch mix of the bench's timed
# loop: tools/profile_summary.py ... last=6 counts those six steps only (not the allocator pre-warm / warm-up), so that per-launch
# averages of size-dependent kernels agree with the live ones
B="--steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-decode"
case $1 in
  base)     bash tools/run/profile_config.sh r4prof_base python3 $R/bench.py $B ;;
  b128)     bash tools/run/profile_config.sh r4prof_b128 python3 $R/bench.py --batch 128 $B ;;
  large)    bash tools/run/profile_config.sh r4prof_large python3 $R/bench.py --model large $B ;;
  large128) bash tools/run/profile_config.sh r4prof_large128 python3 $R/bench.py --model large --batch 128 $B ;;
  decode)   CAL=$(python3 -c "import json;d=json.load(open('$R/profiles/decode_calibration.json'));print('--scale',d['logit_scale'],'--blank-bias',d['blank_bias'])")
            bash tools/run/profile_config.sh r4prof_decode python3 $R/bench_decode.py --decoder beam --streams 2000 --ticks 30 --warmup 5 --from-audio $CAL ;;
  *) echo "unknown configuration $1"; exit 2 ;;
esac
