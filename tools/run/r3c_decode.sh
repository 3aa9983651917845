#!/bin/bash
# round 3: fused LSTM step GEMM for the streaming decoders; bf16-storage oracle checks; smoke
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_decode.py -x -q > $O/t_decode.log 2>&1; echo "decode tests rc=$?"; tail -5 $O/t_decode.log
timeout -k 10 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 $O/smoke.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_train_step.py tests/test_gpu_fullsize.py -x -q -k "oracle or storage" > $O/t_oracle.log 2>&1; echo "oracle tests rc=$?"; tail -8 $O/t_oracle.log
for f in 1 0 1 0; do
  CAIMAN_DECODE_FUSED_LSTM=$f timeout -k 10 300 python3 bench_decode.py --decoder beam --streams 2000 --ticks 60 --warmup 10 --from-audio --scale 5562.699766687201 --blank-bias 466.94 > $O/beam_f${f}_$RANDOM.log 2>&1; echo "beam fused=$f rc=$?"
done
for f in 1 0; do
  CAIMAN_DECODE_FUSED_LSTM=$f timeout -k 10 300 python3 bench_decode.py --decoder greedy --streams 16000 --ticks 40 --warmup 10 --from-audio --no-calibrate > $O/greedy_f${f}.log 2>&1; echo "greedy fused=$f rc=$?"
done
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3c/*.log")):
    for ln in open(f):
        if ln.startswith("{"):
            d = json.loads(ln)
            if "tick_latency_ms" in d:
                print(f, d["streams"], d["tick_latency_ms"], d.get("tokens_per_encoder_frame"), d.get("expansion_rounds_per_tick"))
PY
