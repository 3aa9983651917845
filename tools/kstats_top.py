"""Top kernels of a rocprofv3 --kernel-trace --stats run: python tools/kstats_top.py <kernel_stats.csv> <steps traced> [n]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"<.*$", "", name)
    return name.replace("caiman::", "").replace("(anonymous namespace)::", "")[:60]


def main(path, steps, n=25):
    rows = list(csv.DictReader(open(path)))
    agg = {}
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0.0, 0])
        a[0] += float(r["TotalDurationNs"])
        a[1] += int(r["Calls"])
    tot = sum(a[0] for a in agg.values())
    print(f"total kernel time {tot / steps / 1e6:.2f} ms/step, {sum(a[1] for a in agg.values()) / steps:.0f} launches/step")
    for k, (ns, calls) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:n]:
        print(f"{ns / steps / 1e6:8.3f} ms/step {calls / steps:8.1f} calls/step {ns / calls / 1e3:9.1f} us avg  {100 * ns / tot:5.1f}%  {k}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 25)
