#!/usr/bin/env python3
"""Matrix-pipe occupancy per kernel from one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES pass of bench.py.
SQ_VALU_MFMA_BUSY_CYCLES sums, over all SIMDs, the cycles an MFMA occupies its pipe; GRBM_GUI_ACTIVE sums the busy cycles of the 8 XCDs.
mfma_busy_frac = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs): the fraction of SIMD-cycles with the matrix pipe occupied.
usage: summarize_pmc_mfma.py <pmc_dir> <trace_dir> <steps> <out_json>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def newest(d, pat):
    return max(glob.glob(os.path.join(d, "**", pat), recursive=True), key=os.path.getmtime)


def base(name):
    n = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return n.split("(")[0].strip()


def main():
    pmc, trace, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    per = defaultdict(list)
    for r in csv.DictReader(open(newest(trace, "*kernel_trace.csv"))):
        per[base(r["Kernel_Name"])].append(int(r["Start_Timestamp"]))
    marks = sorted(per[[k for k in per if k.startswith("preprocess_u8_kernel")][0]])
    n_timed = {k: sum(1 for t in v if t >= marks[-steps]) for k, v in per.items()}
    agg = defaultdict(lambda: defaultdict(dict))
    for r in csv.DictReader(open(newest(pmc, "*counter_collection.csv"))):
        k, i = base(r["Kernel_Name"]), int(r["Dispatch_Id"])
        d = agg[k][r["Counter_Name"]]
        d[i] = d.get(i, 0.0) + float(r["Counter_Value"])
    res = {}
    for k, cs in agg.items():
        n = n_timed.get(k, 0)
        if not n or "SQ_VALU_MFMA_BUSY_CYCLES" not in cs:
            continue
        sel = lambda c: [cs[c][i] for i in sorted(cs[c])][-n:]
        busy, gui = sum(sel("SQ_VALU_MFMA_BUSY_CYCLES")), sum(sel("GRBM_GUI_ACTIVE"))
        if busy <= 0 or gui <= 0:
            continue
        res[k] = {"launches_per_step": n / steps, "mfma_busy_frac": round(busy / (gui / 8.0 * 1024.0), 4),
                  "avg_cycles_per_launch": round(gui / 8.0 / n)}
    json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES (own pass of bench.py, timed region only)",
               "kernels": res}, open(out, "w"), indent=1)
    for k, e in sorted(res.items(), key=lambda kv: -kv[1]["avg_cycles_per_launch"] * kv[1]["launches_per_step"]):
        print(f"{k:60s} {e['launches_per_step']:5.1f}/step  mfma_busy {e['mfma_busy_frac']:.3f}")


if __name__ == "__main__":
    main()
