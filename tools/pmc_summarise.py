"""Per-kernel averages of one rocprofv3 --pmc pass.  python tools/pmc_summarise.py <dir> <out.json>
Reads every *counter_collection.csv under <dir>; kernels are grouped by a short name."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(lstm_bwd_resident2_bt|lstm_fwd_resident_bt|lstm_bwd_resident2|lstm_bwd_resident|lstm_fwd_resident|lstm_bwd_step_mfma|lstm_fwd_step_mfma|"
                  r"loss_bwd_colsum_kernel|loss_row_desc_kernel|loss_bwd_kernel|loss_fwd_kernel|lse_rows_kernel|joint_bwd_kernel|"
                  r"joint_fwd_kernel|lamb_stage1|lamb_stage2|beam_topk_kernel|lstm_cell_kernel|proj_gemm_kernel|lstm_images_kernel)", name)
    if m:
        return m.group(1)
    if name.startswith("Cijk") or name.startswith("Custom_Cijk"):
        return "library_gemm"
    return None


def main(root, out):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                if k is None:
                    continue
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    res = {k: {c: {"avg": v[0] / v[1], "dispatches": v[1]} for c, v in d.items()} for k, d in acc.items()}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res)[:1500])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
