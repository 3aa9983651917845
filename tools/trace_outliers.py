"""trace_outliers.py <kernel_trace.csv>: the longest idle gaps of the device timeline and the launches that took far longer
than their kernel's median -- for finding a one-off stall inside a bench run (rocprofv3 --kernel-trace CSV)."""
import csv
import statistics
import sys
from collections import defaultdict


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:90]))
    rows.sort()
    t0 = rows[0][0]
    busy_end, last = rows[0][0], None
    gaps = []
    for s, e, n in rows:
        if s > busy_end:
            gaps.append(((s - busy_end) / 1e3, (busy_end - t0) / 1e6, last, n))
        if e > busy_end:
            busy_end, last = e, n
    print("longest idle gaps (us, at ms, kernel before -> kernel after):")
    for g in sorted(gaps, reverse=True)[:12]:
        print(f"  {g[0]:10.1f} us at {g[1]:10.2f} ms  {g[2]}  ->  {g[3]}")
    by = defaultdict(list)
    for s, e, n in rows:
        by[n].append(((e - s) / 1e3, (s - t0) / 1e6))
    out = []
    for n, v in by.items():
        if len(v) < 4:
            continue
        med = statistics.median(d for d, _ in v)
        for d, at in v:
            if d > 3 * med and d - med > 2000:
                out.append((d - med, d, med, at, n))
    print("launches more than 2 ms and 3x above their kernel's median (excess us, us, median us, at ms):")
    for x in sorted(out, reverse=True)[:12]:
        print(f"  {x[0]:10.1f} {x[1]:10.1f} {x[2]:9.1f} at {x[3]:10.2f} ms  {x[4]}")


if __name__ == "__main__":
    main()
