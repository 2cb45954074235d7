#!/usr/bin/env python3
"""Per-kernel HBM traffic of the bench step, from three rocprofv3 runs of the SAME bench.py command:

  trace_dir : --kernel-trace --stats          (durations; defines the timed region and each kernel's launches per step)
  fetch_dir : --pmc FETCH_SIZE                 (separate pass: FETCH_SIZE and WRITE_SIZE cannot share one on gfx950,
  write_dir : --pmc WRITE_SIZE                  MI355X_MICROARCH.md "rocprofv3 PMC slots")

Units / corrections exactly as MI355X_MICROARCH.md "HBM" prescribes: both counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads (16 B per lane -- what every kernel here issues, LDS-DMA included), so the read
side is doubled; WRITE_SIZE is exact.  Infinity-Cache hits are counted: "traffic" = bytes that left the L2, not DRAM bytes.

For every kernel with >= 1 % of the step (and the byte kernels the north star names: ROIAlign, NMS, paste) the output holds
launches per step, average duration, fetched / written bytes per launch and the rate they imply; for the kernels whose
ALGORITHMIC bytes are a closed form of the bench configuration those are given too, with the over-fetch ratio.

usage: summarize_pmc_traffic.py <trace_dir> <fetch_dir> <write_dir> <bench_json> <steps> <out_json>
"""
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
    n = re.sub(r"^void ", "", name)
    n = n.replace("(anonymous namespace)::", "")
    return n.split("(")[0].strip()


def per_kernel_counter(d, counter):
    """kernel base name -> list of per-dispatch values in dispatch order (summed over the XCD / SE instances of the counter)."""
    agg = defaultdict(dict)
    for r in csv.DictReader(open(newest(d, "*counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            k, i = base(r["Kernel_Name"]), int(r["Dispatch_Id"])
            agg[k][i] = agg[k].get(i, 0.0) + float(r["Counter_Value"])
    return {k: [v[i] for i in sorted(v)] for k, v in agg.items()}


def main():
    trace_dir, fetch_dir, write_dir, bench_json, steps, out = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6]
    bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
    per = defaultdict(list)
    for r in csv.DictReader(open(newest(trace_dir, "*kernel_trace.csv"))):
        per[base(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    marks = sorted(t for t, _ in per[[k for k in per if k.startswith("preprocess_u8_kernel")][0]])
    t_begin = marks[-steps]
    timed = {k: [x for x in v if x[0] >= t_begin] for k, v in per.items()}
    timed = {k: v for k, v in timed.items() if v}
    grand = sum(sum(x[1] for x in v) for v in timed.values())
    fetch, write = per_kernel_counter(fetch_dir, "FETCH_SIZE"), per_kernel_counter(write_dir, "WRITE_SIZE")
    cfg = bench["config"]  # (bench.py's line, or tools/train_bench.py's: images_per_gpu instead of frames_per_step_per_gpu)
    B = cfg.get("frames_per_step_per_gpu", cfg.get("images_per_gpu", 1))
    R = cfg.get("proposals_per_frame", 1000)
    D = cfg.get("raw_detections_per_frame", 0)
    # closed-form algorithmic bytes per launch where the bench configuration fixes them
    algo = {
        "roi_align_fpn_kernel": {"note": "largest launch = the 7x7 box pooler: every live proposal writes 49 bins x 256 channels x 4 B and reads its "
                                         "distinct pyramid cells (<= 26.1 MB per frame, p2-p5)",
                                 "write_bytes_box_pooler": int(B * R * 49 * 256 * 4), "read_bytes_upper_bound_box_pooler": int(B * 26.1e6)},
        "paste_lsq_kernel": {"note": "per kept detection 3.1 KB of mask probabilities + its box window of the 1.2 MB depth map; masks are not "
                                     "materialised in the bench (want_masks False)", "read_bytes": int(B * (D * 3136 + 480 * 640 * 4))},
        "group_nms_kernel": {"note": "latency-bound LDS kernel: <= 1024 boxes x 16 B per group, 128 KiB of suppression words stay in LDS",
                             "read_bytes": int(B * 5 * 1024 * 24)},
    }
    algo_of = lambda k: next((v for a, v in algo.items() if k == a or k.startswith(a + "<")), None)  # (template arguments stay in the base name)
    want = {k for k, v in timed.items() if sum(x[1] for x in v) >= 0.01 * grand} | {k for k in timed if algo_of(k)}
    res = {}
    for k in sorted(want, key=lambda k: -sum(x[1] for x in timed[k])):
        n = len(timed[k])
        dur = sum(x[1] for x in timed[k]) / n * 1e-9
        f = fetch.get(k, [])[-n:]
        w = write.get(k, [])[-n:]
        fb = 2.0 * 1024.0 * sum(f) / max(len(f), 1)
        wb = 1024.0 * sum(w) / max(len(w), 1)
        e = {"launches_per_step": n / steps, "avg_launch_us": round(dur * 1e6, 2), "share_of_step": round(sum(x[1] for x in timed[k]) / grand, 4),
             "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb), "hbm_bytes_per_launch": round(fb + wb),
             "achieved_GBps": round((fb + wb) / dur / 1e9, 1), "counter_launches_averaged": min(len(f), len(w))}
        if algo_of(k):
            e["algorithmic"] = algo_of(k)
        res[k] = e
    ra = next((v for k, v in res.items() if k == "roi_align_fpn_kernel" or k.startswith("roi_align_fpn_kernel<")), None)
    if ra and "algorithmic" in ra:
        # the box pooler is the first and by far largest of the step's three launches: weigh the per-launch averages back to it
        tot_f = ra["fetch_bytes_per_launch"] * ra["launches_per_step"]
        tot_w = ra["write_bytes_per_launch"] * ra["launches_per_step"]
        a = ra["algorithmic"]
        ra["per_step_fetch_bytes"], ra["per_step_write_bytes"] = round(tot_f), round(tot_w)
        ra["over_fetch_vs_distinct_cells"] = round(tot_f / a["read_bytes_upper_bound_box_pooler"], 2)
    doc = {"command": sys.argv[7] if len(sys.argv) > 7 else "python3 bench.py --steps %d --warmup 3 --no-cpu-baseline --no-alt-modes --no-operating-points" % steps,
           "source": "rocprofv3 --kernel-trace (durations), --pmc FETCH_SIZE, --pmc WRITE_SIZE: three separate passes; FETCH_SIZE x2 "
                     "(gfx950 wide-read correction), KiB -> bytes; Infinity-Cache hits are counted as traffic",
           "bench_value": bench["value"], "ms_per_step": bench["ms_per_step"], "kernels": res}
    json.dump(doc, open(out, "w"), indent=1)
    for k, e in res.items():
        print(f"{k:60s} {e['launches_per_step']:5.1f}/step {e['avg_launch_us']:9.1f} us  fetch {e['fetch_bytes_per_launch'] / 1e6:9.1f} MB  "
              f"write {e['write_bytes_per_launch'] / 1e6:9.1f} MB  {e['achieved_GBps']:8.1f} GB/s")


if __name__ == "__main__":
    main()
