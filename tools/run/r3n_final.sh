#!/bin/bash
# final tree: full GPU suite, smoke, default bench line (with decode + cpu baseline), variants, phase timers
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${RUN_TAG:-r3n}
mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $O/t_all.log 2>&1; echo "ALL gpu tests rc=$?"; tail -3 $O/t_all.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 600 python3 bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "default bench rc=$?"
timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode --feed > $O/bench_feed.json 2> $O/bench_feed.err; echo "feed rc=$?"
timeout -k 10 300 python3 bench.py --model large --steps 8 --warmup 2 --no-decode > $O/large.json 2> $O/large.err; echo "large rc=$?"
timeout -k 10 300 python3 bench.py --batch 128 --steps 6 --warmup 2 --no-cpu-baseline --no-decode > $O/b128.json 2> $O/b128.err; echo "b128 rc=$?"
timeout -k 10 300 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-decode > $O/gpus2.json 2> $O/gpus2.err; echo "gpus2 rehearsal rc=$?"
timeout -k 10 300 python3 tools/lstm_resident_bench.py --skip-agreement > $O/phase_1024.log 2>&1; echo "phase 1024 rc=$?"
timeout -k 10 300 python3 tools/lstm_resident_bench.py --skip-agreement --hidden 1536 --layers 5 > $O/phase_1536.log 2>&1; echo "phase 1536 rc=$?"
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/" + __import__("os").environ.get("RUN_TAG", "r3n") + "/*.json")):
    try:
        d = json.load(open(f))
        print(f, round(d["ms_per_step"], 2), round(d["value"], 3), d.get("kernel_ms_per_step"), d["lstm_resident"], d.get("feed"), (d.get("decode") or {}).get("tick_latency_ms"), d.get("rehearsal"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $O/phase_1024.log | cut -c1-900; tail -3 $O/phase_1536.log | cut -c1-900
