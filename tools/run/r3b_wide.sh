#!/bin/bash
# round 3: H = 1536 resident kernels (DMA forward, 3-stage 2-D split backward), reducer fixes, world-1 nccl test
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_resident_oracle.py tests/test_gpu_lstm.py -x -q > $O/t_lstm.log 2>&1; echo "lstm tests rc=$?"; tail -5 $O/t_lstm.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_distributed.py -x -q > $O/t_dist.log 2>&1; echo "dist tests rc=$?"; tail -5 $O/t_dist.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_train_step.py tests/test_gpu_fullsize.py -x -q > $O/t_step.log 2>&1; echo "step tests rc=$?"; tail -5 $O/t_step.log
for kb in 1 2; do
  CAIMAN_LSTM_WIDE_KB=$kb timeout -k 10 300 python3 bench.py --model large --steps 8 --warmup 2 --no-decode > $O/large_kb$kb.json 2> $O/large_kb$kb.err; echo "large kb$kb rc=$?"
done
for dma in 0 1 0 1; do
  CAIMAN_LSTM_FWD_DMA=$dma timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/base_dma${dma}_$RANDOM.json 2> $O/base_dma$dma.err; echo "base dma$dma rc=$?"
done
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3b/*.json")):
    try:
        d = json.load(open(f))
        print(f, round(d["ms_per_step"], 2), round(d["value"], 3), d.get("kernel_ms_per_step", {}).get("lstm_fwd"), d.get("kernel_ms_per_step", {}).get("lstm_bwd"), d["host_ms_per_step"], d["lstm_resident"])
    except Exception as e:
        print(f, "ERR", e)
PY
