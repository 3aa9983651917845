#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3i
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_resident_oracle.py tests/test_gpu_lstm.py -x -q > $O/t_lstm.log 2>&1; echo "lstm tests rc=$?"; tail -3 $O/t_lstm.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_train_step.py tests/test_gpu_fullsize.py tests/test_gpu_model.py -x -q > $O/t_step.log 2>&1; echo "step tests rc=$?"; tail -3 $O/t_step.log
for i in 1 2; do
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/base_$i.json 2> $O/base.err; echo "base rc=$?"
done
timeout -k 10 300 python3 bench.py --model large --steps 8 --warmup 2 --no-decode > $O/large.json 2> $O/large.err; echo "large rc=$?"
timeout -k 10 300 python3 bench.py --batch 128 --steps 6 --warmup 2 --no-cpu-baseline --no-decode > $O/b128.json 2> $O/b128.err; echo "b128 rc=$?"
timeout -k 10 300 python3 tools/lstm_resident_bench.py > $O/phase.log 2>&1; echo "phase timers rc=$?"; tail -25 $O/phase.log
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3i/*.json")):
    try:
        d = json.load(open(f))
        print(f, round(d["ms_per_step"], 2), round(d["value"], 3), d.get("kernel_ms_per_step"), d["lstm_resident"], d.get("roofline", {}).get("chain"))
    except Exception as e:
        print(f, "ERR", e)
PY
