"""profiles/<tag>_{kernel_stats.csv, pmc_traffic.json, summary.md} from the rocprofv3 passes of ONE command
(tools/run/profile_config.sh runs them: a kernel-trace pass and four one-counter PMC passes, all with IDENTICAL program
arguments, so that every pass sees the same launches).

    python tools/profile_summary.py <dir with stats/ FETCH_SIZE/ WRITE_SIZE/ SQ_VALU_MFMA_BUSY_CYCLES/ GRBM_GUI_ACTIVE/> <tag> <steps traced> ["title"]

Round-2 verdict, "profile hygiene": a byte count from one launch population divided by a time from another is not a
rate.  Here a kernel's `traffic_bytes_per_launch` and `kernel_avg_us` are only combined into `hbm_side_rate_gbs` when the
kernel was launched the same number of times in the kernel-trace pass and in every counter pass (`same_population`);
the times come from the kernel-trace pass (kernels run serialised and slower under counter collection).
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

NAMES = ("lstm_bwd_resident2_bt_dma|lstm_bwd_resident2_bt|lstm_fwd_resident_bt_dma|lstm_fwd_resident_bt|lstm_fwd_resident_dma|lstm_bwd_resident2|lstm_bwd_resident|lstm_fwd_resident|"
         "lstm_bwd_step_mfma|lstm_fwd_step_mfma|loss_bwd_colsum_kernel|loss_row_desc_kernel|loss_bwd_kernel|loss_fwd_kernel|lse_rows_kernel|"
         "lse_partials_kernel|joint_fc_gemm8_kernel|joint_fc_gemm_kernel|joint_wgrad8_kernel|joint_wgrad_kernel|occupy_cus_kernel|joint_bwd_kernel|joint_fwd_kernel|lamb_stage1|lamb_stage2|gnorm_kernel|beam_topk_reg_kernel|beam_topk_kernel|"
         "lstm_cell_kernel|gather_inputs_kernel|joint_act_kernel|proj_gemm_kernel|lstm_images_kernel|lstm_grad_deliver_kernel|logmel_kernel|"
         "mel_normalize_kernel|specaug_splice_kernel|dbias_rows_kernel|"
         # round 4, launch sweep: this repo's small kernels by name (until then they sat in "other" with the torch glue)
         "embedding_grad_kernel|slab_accumulate_kernel|lstm_last_states_kernel|specaug_geometry_kernel|gnorm_partial_kernel|"
         "gnorm_final_kernel|lamb_ratio_kernel|tile_rows_kernel|dropout_mask_kernel")


def short(name):
    m = re.search("(" + NAMES + ")", name)
    if m:
        k = m.group(1)
        if k == "proj_gemm_kernel" and "Lb1EEEv" in name:   # the CELL instantiation (last template argument true): LSTM step GEMM
            return "proj_gemm_kernel[cell]"
        return k
    if name.startswith("Cijk") or name.startswith("Custom_Cijk"):
        return "library_gemm"
    if "rccl" in name.lower() or "nccl" in name.lower():
        return "rccl"
    return None


STEP_END = "lamb_stage2"     # the last kernel of a training step: `last=N` keeps the dispatches of the final N steps


def _last_steps(names, last):
    """index range [lo, hi) of the dispatches (in launch order) that make up the final `last` steps, or None"""
    ends = [i for i, n in enumerate(names) if STEP_END in n]
    if last is None or len(ends) < last + 1:
        return None
    return ends[-last - 1] + 1, ends[-1] + 1


def counter_pass(root, counter, last=None):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{root}/{counter}/**/*counter_collection.csv", recursive=True):
        per = {}       # dispatch id -> [kernel name, summed value]: a counter may come as several rows per dispatch
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                e = per.setdefault(int(row["Dispatch_Id"]), [row["Kernel_Name"], 0.0])
                e[1] += float(row["Counter_Value"])
        ids = sorted(per)
        rng = _last_steps([per[i][0] for i in ids], last)
        if rng:
            ids = ids[rng[0]:rng[1]]
        for i in ids:
            k = short(per[i][0])
            if k is None:
                continue
            acc[k][0] += per[i][1]
            acc[k][1] += 1
    return {k: {"avg": v[0] / v[1], "dispatches": v[1]} for k, v in acc.items()}


def main(root, tag, steps, title=None, last=None):
    """last = N (sixth argument `last=N`): only the final N training steps of the traced process count (delimited by the
    optimiser's last kernel) -- the timed steps of a bench run, without its allocator pre-warm and warm-up steps, so that
    per-launch averages of size-dependent kernels are those of the bench's own batch mix."""
    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    last = int(str(last).split("=")[-1]) if last else None
    if steps == "auto":   # the program's own JSON line in the kernel-trace pass's log says how many steps / ticks it ran
        steps = None
        for ln in open(f"{root}/stats.log"):
            if ln.lstrip().startswith("{"):
                d = json.loads(ln)
                steps = d.get("steps_executed_in_process") or (d.get("ticks", 0) + d.get("warmup_ticks", 0)) or None
        assert steps, "no step count in stats.log: give it on the command line"
    steps = float(steps)
    stats_csv = glob.glob(f"{root}/stats/**/*_kernel_stats.csv", recursive=True)[0]
    agg = defaultdict(lambda: [0.0, 0])
    cut = None
    if last:
        trace = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"])
                        for r in csv.DictReader(open(glob.glob(f"{root}/stats/**/*_kernel_trace.csv", recursive=True)[0]))))
        cut = _last_steps([t[2] for t in trace], last)
    if cut:
        byname = defaultdict(lambda: [0, 0.0])
        for s0, e0, n in trace[cut[0]:cut[1]]:
            byname[n][0] += 1
            byname[n][1] += e0 - s0
            k = short(n) or "other"
            agg[k][0] += e0 - s0
            agg[k][1] += 1
        with open(f"{prof}/{tag}_kernel_stats.csv", "w", newline="") as fh:     # the same columns as rocprofv3's own summary
            wr = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
            wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
            total = sum(v[1] for v in byname.values())
            for n, (c, ns) in sorted(byname.items(), key=lambda kv: -kv[1][1]):
                wr.writerow([n, c, int(ns), round(ns / c, 1), round(100.0 * ns / total, 3)])
        steps = float(last)
    else:
        shutil.copy(stats_csv, f"{prof}/{tag}_kernel_stats.csv")
        for r in csv.DictReader(open(stats_csv)):
            k = short(r["Name"]) or "other"
            agg[k][0] += float(r["TotalDurationNs"])
            agg[k][1] += int(r["Calls"])
    counters = ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")
    S = {c: counter_pass(root, c, last if cut else None) for c in counters}
    pm = {"source": "rocprofv3 --kernel-trace --stats (times) and rocprofv3 --pmc <counter> --kernel-trace, ONE counter per pass, all passes "
                    "with identical program arguments (tools/run/profile_config.sh); aggregated by tools/profile_summary.py",
          "unit": "traffic_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reports half of wide coalesced reads, "
                  "MI355X_MICROARCH.md HBM section); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs); "
                  "hbm_side_rate_gbs = traffic_bytes_per_launch / kernel_avg_us, only where same_population"}
    for k in sorted(set(S["FETCH_SIZE"]) | set(agg)):
        if k == "other":
            continue
        f = S["FETCH_SIZE"].get(k)
        e = {}
        if k in agg:
            e["launches_kernel_trace_pass"] = agg[k][1]
            e["kernel_avg_us"] = round(agg[k][0] / agg[k][1] / 1e3, 2)
            e["ms_per_step"] = round(agg[k][0] / steps / 1e6, 3)
        if f:
            w = S["WRITE_SIZE"].get(k, {"avg": 0.0, "dispatches": 0})
            m = S["SQ_VALU_MFMA_BUSY_CYCLES"].get(k, {"avg": 0.0, "dispatches": 0})
            g = S["GRBM_GUI_ACTIVE"].get(k, {"avg": 0.0, "dispatches": 0})
            counts = {f["dispatches"], w["dispatches"], m["dispatches"], g["dispatches"]}
            e.update({"launches_counter_passes": sorted(counts), "fetch_size_kib_avg": round(f["avg"], 1),
                      "write_size_kib_avg": round(w["avg"], 1), "traffic_bytes_per_launch": int((2 * f["avg"] + w["avg"]) * 1024),
                      "mfma_util": round(m["avg"] / (1024 * g["avg"] / 8), 4) if g["avg"] else None})
            same = k in agg and counts == {agg[k][1]}
            e["same_population"] = same
            if same:
                e["hbm_side_rate_gbs"] = round(e["traffic_bytes_per_launch"] / (e["kernel_avg_us"] * 1e-6) / 1e9, 1)
        pm[k] = e
    json.dump(pm, open(f"{prof}/{tag}_pmc_traffic.json", "w"), indent=1)

    tot = sum(v[0] for v in agg.values()) / steps / 1e6
    launches = sum(v[1] for v in agg.values()) / steps
    L = [f"# {title or tag} — rocprofv3\n",
         f"Kernel-trace pass: {launches:.0f} launches and {tot:.2f} ms of kernel time per step / tick ({steps:.0f} "
         f"{'timed steps = the last ones of the process' if cut else 'traced'}; `{tag}_kernel_stats.csv`). "
         f"Counter passes: one counter each, same arguments (`{tag}_pmc_traffic.json`).\n",
         "| kernel | launches / step | ms / step | avg us | % of kernel time | HBM-side traffic / launch | MFMA busy | HBM-side rate |",
         "|---|---|---|---|---|---|---|---|"]
    for k, (ns, calls) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:24]:
        e = pm.get(k, {}) if k != "other" else {}
        tr = f"{e['traffic_bytes_per_launch'] / 1e6:.1f} MB" if "traffic_bytes_per_launch" in e else ""
        mf = f"{100 * e['mfma_util']:.1f} %" if e.get("mfma_util") is not None else ""
        rate = f"{e['hbm_side_rate_gbs'] / 1e3:.2f} TB/s" if "hbm_side_rate_gbs" in e else ("(populations differ)" if tr else "")
        L.append(f"| `{k}` | {calls / steps:.1f} | {ns / steps / 1e6:.3f} | {ns / calls / 1e3:.1f} | {100 * ns / (tot * steps * 1e6):.1f} | {tr} | {mf} | {rate} |")
    open(f"{prof}/{tag}_summary.md", "w").write("\n".join(L) + "\n")
    print("\n".join(L))


if __name__ == "__main__":
    main(*sys.argv[1:6])
