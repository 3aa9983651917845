#!/bin/bash
# r4_graph.sh <out>: the beam round as one captured graph launch (CAIMAN_BEAM_GRAPH=1): the decode tests under the switch, then
# bench_decode.py at 2 000 streams with the switch off / on, alternating, on one box.
set -o pipefail
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_decode.py -m gpu -x -q > $O/tests.log 2>&1 || { echo "tests failed"; tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
CAL=$(python3 -c "import json;d=json.load(open('profiles/decode_calibration.json'));print('--scale',d['logit_scale'],'--blank-bias',d['blank_bias'])")
for r in 1 2; do
  for v in 0 1; do
    CAIMAN_BEAM_GRAPH=$v timeout -k 10 300 python bench_decode.py --decoder beam --streams ${STREAMS:-2000} --ticks 100 --warmup 10 --from-audio $CAL > $O/graph${v}_$r.json 2> $O/graph${v}_$r.err || { echo "graph=$v failed"; tail -5 $O/graph${v}_$r.err; exit 1; }
    python3 -c "
import json
d=json.loads(open('$O/graph${v}_$r.json').read().strip().splitlines()[-1])
print('graph=$v round $r:', d['tick_latency_ms'], d['real_time'], d.get('tokens_per_encoder_frame'))"
  done
done
