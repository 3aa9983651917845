#!/bin/bash
# stall_probe.sh [extra bench args]: twelve short bench runs on one box, alternating kernel timing on / off; prints each run's
# mean, first and longest step on the device clock (a one-off stall shows as a step of 50 ms instead of 36).
cat > /tmp/stall_line.py <<'PY'
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d["step_ms_device"]
print(sys.argv[1], round(d["ms_per_step"], 2), "device-clock mean", round(sum(s) / len(s), 2), "first step", s[0], "max step", max(s),
      "host issue first/mean", d["host_issue_ms"][0], round(sum(d["host_issue_ms"]) / len(s), 2), d["host_ms_per_step"], "STALL" if max(s) > 42 else "")
PY
for i in 1 2 3 4 5 6; do
  for t in "" "--no-kernel-timing"; do
    timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-decode $t "$@" 2>/dev/null | python3 /tmp/stall_line.py "timing_${t:-on}" || exit 1
  done
done
