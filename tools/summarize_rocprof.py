#!/usr/bin/env python3
"""Turns a `rocprofv3 --kernel-trace --stats --output-format csv` run of bench.py into the summary committed
under profiles/:
  * the whole-process per-kernel statistics exactly as rocprofv3 reports them (*_kernel_stats.csv), and
  * the same averages restricted to the TIMED steps of bench.py (the last `steps` batches).  Every step starts
    with exactly one `preprocess_u8_kernel` dispatch, so the timed region is everything that starts at or after
    the `steps`-th last dispatch of that kernel in *_kernel_trace.csv -- calibration and warm-up launches
    (different shapes) are thereby excluded and the dominant kernel's average can be compared with the HIP-event
    figure bench.py prints in `roofline.avg_launch_ms`.
usage: summarize_rocprof.py <rocprof_out_dir> <bench_json> <steps> <warmup> <out_md>
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d, bench_json, steps, warmup, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    command = sys.argv[6] if len(sys.argv) > 6 else (f"python3 bench.py --steps {steps} --warmup {warmup} --no-cpu-baseline --no-alt-modes --no-operating-points --no-train-leg")
    stats = max(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    trace = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
    rows = list(csv.DictReader(open(stats)))
    per = defaultdict(list)
    for r in csv.DictReader(open(trace)):
        per[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    marks = sorted(t for t, _ in per[[k for k in per if k.startswith("preprocess_u8_kernel")][0]])
    t_begin = marks[-steps]
    lines = ["# rocprofv3 kernel summary", "",
             f"command: `rocprofv3 --kernel-trace --stats --output-format csv -- {command}`", "",
             f"bench line of the profiled run: value={bench['value']} {bench['unit']}, ms_per_step={bench['ms_per_step']}, "
             f"roofline={json.dumps({k: bench['roofline'][k] for k in ('kernel', 'achieved', 'peak', 'frac', 'launches', 'avg_launch_ms')})}", "",
             "## timed region only (last %d steps), per kernel" % steps, "",
             "| kernel | launches/step | avg us | total ms/step | share |", "|---|---|---|---|---|"]
    timed = []
    for name, lst in per.items():
        sel = [x for x in lst if x[0] >= t_begin]
        if not sel:
            continue
        tot = sum(x[1] for x in sel)
        timed.append((tot, name, len(sel) / steps, tot / len(sel) / 1e3, tot / steps / 1e6))
    timed.sort(reverse=True)
    grand = sum(t[0] for t in timed)
    for tot, name, lps, avg_us, ms_step in timed[:25]:
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append(f"| `{short}` | {lps:g} | {avg_us:.1f} | {ms_step:.3f} | {100 * tot / grand:.1f}% |")
    lines += ["", f"sum of kernel time per timed step: {grand / steps / 1e6:.2f} ms (wall ms_per_step {bench['ms_per_step']})", "",
              "## whole process (verbatim rocprofv3 --stats, top 25; includes BN calibration and warm-up launches)", "",
              "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows[:25]:
        name = r["Name"]
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append(f"| `{short}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main()
