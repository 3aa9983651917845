"""trace_gaps.py <kernel_trace.csv> [--steps N]: where the device sits idle inside a training step.

Reads a rocprofv3 --kernel-trace CSV of `bench.py`, takes the last N steps (a step ends with `lamb_stage2`), merges the
kernels' [start, end) intervals over all queues and lists the idle intervals by the kernel that FOLLOWS them: a gap in
front of a kernel is time the device waited for that launch (host issue, a stream dependency or the launch itself)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"at::native::", "", name)
    return name[:70]


def main():
    path = sys.argv[1]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 3
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "lamb_stage2" in r[2]]
    if len(ends) < steps + 1:
        raise SystemExit(f"only {len(ends)} steps in the trace")
    lo, hi = ends[-steps - 1] + 1, ends[-1] + 1
    rows = rows[lo:hi]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy_end = rows[0][0]
    gaps = defaultdict(lambda: [0, 0.0])
    prev = defaultdict(lambda: [0, 0.0])
    last_name = None
    busy = 0.0
    for s, e, n in rows:
        if s > busy_end:
            g = (s - busy_end) / 1000.0
            a = gaps[short(n)]
            a[0] += 1
            a[1] += g
            b = prev[short(last_name) if last_name else "?"]
            b[0] += 1
            b[1] += g
            busy += (e - s) / 1000.0
            busy_end = e
            last_name = n
        else:
            if e > busy_end:
                busy += (e - busy_end) / 1000.0
                busy_end = e
                last_name = n
    total = (t1 - t0) / 1000.0
    print(f"{steps} steps: {total / steps / 1000:.3f} ms per step on the device clock, busy {busy / steps / 1000:.3f} ms, "
          f"idle {(total - busy) / steps / 1000:.3f} ms")
    print("idle time by the kernel that follows the gap (us per step, gaps per step, mean us):")
    for n, (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {g / steps:8.1f} {c / steps:6.1f} {g / c:7.1f}  {n}")
    print("idle time by the kernel in front of the gap:")
    for n, (c, g) in sorted(prev.items(), key=lambda kv: -kv[1][1])[:15]:
        print(f"  {g / steps:8.1f} {c / steps:6.1f} {g / c:7.1f}  {n}")


if __name__ == "__main__":
    main()
