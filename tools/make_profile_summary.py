"""profiles/rNN_* from the outputs of the rocprofv3 passes (run on the GPU box, outputs merged back under gpurun_out/):

    rocprofv3 --kernel-trace --stats --output-format csv -d <dir>/stats -- python3 bench.py --steps 6 --warmup 3 \
        --no-cpu-baseline --no-kernel-timing --no-decode
    for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE:      # ONE counter per pass
        rocprofv3 --pmc $c --kernel-trace --output-format csv -d <dir>/$c -- python3 bench.py --steps 2 --warmup 1 ...
        python3 tools/pmc_summarise.py <dir>/$c <dir>/sum_$c.json

    python tools/make_profile_summary.py <dir> <bench line .json> <round tag, e.g. r02>
"""
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summarise import short  # noqa: E402


def main(root, bench_json, tag):
    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    stats_csv = glob.glob(f"{root}/stats/*/*_kernel_stats.csv")[0]
    shutil.copy(stats_csv, f"{prof}/{tag}_bench_kernel_stats.csv")
    shutil.copy(bench_json, f"{prof}/{tag}_bench_final.json")
    rows = list(csv.DictReader(open(stats_csv)))
    n = 11   # 2 allocator pre-warm + 3 warm-up + 6 timed steps
    S = {c: json.load(open(f"{root}/sum_{c}.json")) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")}
    avg = {}
    for r in rows:
        k = short(r["Name"])
        if k:
            a = avg.setdefault(k, [0.0, 0])
            a[0] += float(r["TotalDurationNs"])
            a[1] += int(r["Calls"])
    pm = {"source": "rocprofv3 --pmc <counter> --kernel-trace, ONE counter per pass (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES, "
                    "GRBM_GUI_ACTIVE) on `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-decode`; "
                    "aggregated by tools/pmc_summarise.py + tools/make_profile_summary.py",
          "unit": "traffic bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE reports half of wide coalesced reads: "
                  "MI355X_MICROARCH.md, HBM section); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs).  Kernels run "
                  "serialised and slower under counter collection: use the ratios, not the cycle counts, as times."}
    for k in sorted(S["FETCH_SIZE"]):
        f = S["FETCH_SIZE"][k]["FETCH_SIZE"]
        w = S["WRITE_SIZE"].get(k, {}).get("WRITE_SIZE", {"avg": 0.0})
        m = S["SQ_VALU_MFMA_BUSY_CYCLES"].get(k, {}).get("SQ_VALU_MFMA_BUSY_CYCLES", {"avg": 0.0})
        g = S["GRBM_GUI_ACTIVE"].get(k, {}).get("GRBM_GUI_ACTIVE", {"avg": 0.0})
        e = {"launches": f["dispatches"], "fetch_size_kib_avg": round(f["avg"], 1), "write_size_kib_avg": round(w["avg"], 1),
             "traffic_bytes_per_launch": int((2 * f["avg"] + w["avg"]) * 1024), "mfma_busy_cycles_avg": round(m["avg"], 1),
             "gui_active_cycles_avg": round(g["avg"], 1), "mfma_util": round(m["avg"] / (1024 * g["avg"] / 8), 4) if g["avg"] else None}
        if k in avg:
            e["rocprof_kernel_avg_us"] = round(avg[k][0] / avg[k][1] / 1e3, 2)
        pm[k] = e
    json.dump(pm, open(f"{prof}/{tag}_pmc_traffic.json", "w"), indent=1)

    tot = sum(float(r["TotalDurationNs"]) for r in rows) / n / 1e6
    launches = sum(int(r["Calls"]) for r in rows) / n

    def grp(pred):
        return sum(float(r["TotalDurationNs"]) for r in rows if pred(r["Name"])) / n / 1e6

    gemm = grp(lambda s: s.startswith("Cijk") or s.startswith("Custom"))
    lstm = grp(lambda s: "lstm_" in s and "resident" in s)
    proj = grp(lambda s: "proj_gemm" in s)
    loss = grp(lambda s: any(k in s for k in ("loss_", "lse_rows", "joint_fwd", "joint_bwd")))
    opt = grp(lambda s: "lamb_" in s or "gnorm" in s)
    rest = tot - gemm - lstm - proj - loss - opt
    b = json.load(open(bench_json))
    L = [f"# Round {int(tag[1:])}, final tree — rocprofv3\n"]
    L.append("`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing "
             f"--no-decode` (11 steps traced: 2 allocator pre-warm + 3 warm-up + 6 timed; `{tag}_bench_kernel_stats.csv`). {launches:.0f} kernel launches and "
             f"{tot:.1f} ms of kernel time per step under the profiler: library GEMMs {gemm:.1f}, weight-resident LSTM kernels {lstm:.1f}, grouped projection "
             f"GEMM {proj:.1f}, joint + loss kernels {loss:.1f}, optimiser {opt:.1f}, everything else {rest:.1f}.\n")
    d = b.get("decode", {})
    L.append(f"Default `python bench.py` on the same tree (`{tag}_bench_final.json`): **{b['ms_per_step']:.2f} ms/step = {b['value']:.3f} audio-hours/s**; "
             f"host thread: {b['host_ms_per_step']['issue']} ms to queue a step, {b['host_ms_per_step']['wait_for_device']} ms waiting for the device; live "
             f"HIP-event bracket of the dominant kernel {b['roofline']['avg_launch_us']:.0f} us per launch, {b['roofline']['chain']['us_per_timestep']:.2f} us per "
             f"dependent timestep of the longest slot; {b['lstm_resident']['launches']} resident launches, {b['lstm_resident']['handoff_timeouts']} hand-off "
             f"timeouts; decode record: {d.get('streams', '?')} beam streams, tick latency p50 {d.get('tick_latency_ms', {}).get('p50', 0):.1f} / p99 "
             f"{d.get('tick_latency_ms', {}).get('p99', 0):.1f} ms (60 ms budget).\n")
    L.append("| kernel | calls/step | ms / step | avg us | % |\n|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
        ms = float(r["TotalDurationNs"]) / n / 1e6
        L.append(f"| `{r['Name'][:100]}` | {int(r['Calls']) / n:.1f} | {ms:.3f} | {float(r['AverageNs']) / 1e3:.2f} | {100 * ms / tot:.1f} |")
    L.append("")
    L.append(f"PMC (separate `--pmc` passes, one counter each, `{tag}_pmc_traffic.json`; traffic = (2·FETCH_SIZE + WRITE_SIZE)·1024 per the guide's gfx950 correction):\n")
    L.append("| kernel | traffic per launch | MFMA busy | rocprof avg us |\n|---|---|---|---|")
    for k in ("lstm_bwd_resident2", "lstm_fwd_resident", "proj_gemm_kernel", "library_gemm", "loss_bwd_colsum_kernel", "lse_rows_kernel",
              "loss_fwd_kernel", "joint_fwd_kernel", "joint_bwd_kernel"):
        if isinstance(pm.get(k), dict):
            e = pm[k]
            L.append(f"| `{k}` | {e['traffic_bytes_per_launch'] / 1e6:.0f} MB | {100 * (e['mfma_util'] or 0):.1f} % | {e.get('rocprof_kernel_avg_us', '')} |")
    ph_path = f"{prof}/{tag}_lstm_phase_timers.json"
    if os.path.exists(ph_path):
        ph = json.load(open(ph_path))
        s, w = ph["split_backward_kernel (default)"], ph["whole_row_backward_kernel (CAIMAN_LSTM_BWD_SPLIT=0)"]
        L.append(f"\nPhase timers of one workgroup (`tools/lstm_resident_bench.py`, 8 layers x 256 steps, H = 1024, B = 32; `{tag}_lstm_phase_timers.json`), us per timestep:\n")
        L.append("| kernel | phases | sum |\n|---|---|---|")
        f = s["fwd_us_per_timestep"]
        L.append(f"| `lstm_fwd_resident` | wait {f['wait']} + h row into LDS {f['operand_to_lds']} + MFMA and cell {f['mfma_cell']} + drain {f['drain_barrier']} | {sum(f.values()):.2f} |")
        f = s["bwd2_us_per_timestep"]
        L.append(f"| `lstm_bwd_resident2` | wait for the K quarter {f['wait_quarter']} + gather with MFMAs {f['gather_mfma']} + partials out {f['partials_out_drain']} + "
                 f"wait for the group {f['wait_group']} + partials in and epilogue {f['partials_in_epilogue']} + drain {f['drain_barrier']} | {sum(f.values()):.2f} |")
        f = w["bwd_us_per_timestep"]
        L.append(f"| `lstm_bwd_resident` (round 1) | wait {f['wait']} + 256 KB gather with MFMAs {f['operand_to_lds']} + epilogue {f['mfma_cell']} + drain {f['drain_barrier']} | {sum(f.values()):.2f} |")
        L.append(f"\nThe same 8-layer stack, forward + backward: per-timestep launches {s['step_fwd_bwd_ms']:.1f} ms, resident {w['resident_fwd_bwd_ms']:.1f} ms with the "
                 f"whole-row backward kernel, {s['resident_fwd_bwd_ms']:.1f} ms with the 2-D split.\n")
    open(f"{prof}/{tag}_bench_summary.md", "w").write("\n".join(L))
    print("\n".join(L)[:1800])


if __name__ == "__main__":
    main(*sys.argv[1:4])
