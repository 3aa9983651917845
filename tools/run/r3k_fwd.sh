#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3k
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_resident_oracle.py tests/test_gpu_lstm.py -x -q > $O/t_lstm.log 2>&1; echo "lstm tests rc=$?"; tail -3 $O/t_lstm.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_train_step.py tests/test_gpu_model.py -x -q > $O/t_step.log 2>&1; echo "step tests rc=$?"; tail -3 $O/t_step.log
for rep in 1 2; do for v in DEFAULT OLDFWD; do
  if [ $v = DEFAULT ]; then unset CAIMAN_LIB_OVERRIDE; else export CAIMAN_LIB_OVERRIDE=$R/caiman_asr_amd/lib/variants/libcaiman_$v.so; fi
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/base_${v}_$rep.json 2> $O/base.err; echo "$v rc=$?"
done; done
for v in DEFAULT OLDFWD; do
  if [ $v = DEFAULT ]; then unset CAIMAN_LIB_OVERRIDE; else export CAIMAN_LIB_OVERRIDE=$R/caiman_asr_amd/lib/variants/libcaiman_$v.so; fi
  timeout -k 10 300 python3 bench.py --model large --steps 8 --warmup 2 --no-decode > $O/large_${v}.json 2> $O/large.err; echo "large $v rc=$?"
done
unset CAIMAN_LIB_OVERRIDE
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3k/*.json")):
    try:
        d = json.load(open(f))
        print(f, round(d["ms_per_step"], 2), d["kernel_ms_per_step"]["lstm_fwd"], d["kernel_ms_per_step"]["lstm_bwd"], d["lstm_resident"])
    except Exception as e:
        print(f, "ERR", e)
PY
