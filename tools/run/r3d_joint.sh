#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_joint_gemm.py -x -q > $O/t_joint.log 2>&1; echo "joint tests rc=$?"; tail -6 $O/t_joint.log
timeout -k 10 300 python3 tools/joint_gemm_bench.py > $O/joint_bench.json 2> $O/joint_bench.err; echo "joint bench rc=$?"; cat $O/joint_bench.json
timeout -k 10 300 python3 tools/joint_gemm_bench.py --rows 1200000 --rounds 3 > $O/joint_bench_b128.json 2> $O/joint_bench_b128.err; echo "joint bench b128 rc=$?"; cat $O/joint_bench_b128.json
timeout -k 10 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_train_step.py tests/test_gpu_fullsize.py tests/test_gpu_decode.py -x -q -k "oracle or storage or fused or large_batch" > $O/t_oracle.log 2>&1; echo "oracle tests rc=$?"; tail -8 $O/t_oracle.log
for jg in 0 1 fwd 0 1; do
  CAIMAN_JOINT_GEMM=$jg timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $O/base_jg${jg}_$RANDOM.json 2> $O/base_jg$jg.err; echo "base jg=$jg rc=$?"
done
timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode --feed > $O/base_feed.json 2> $O/base_feed.err; echo "feed rc=$?"
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/r3d/base_*.json")):
    try:
        d = json.load(open(f))
        print(f, round(d["ms_per_step"], 2), round(d["value"], 3), d.get("kernel_ms_per_step"), d.get("feed"))
    except Exception as e:
        print(f, "ERR", e)
PY
