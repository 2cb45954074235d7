#!/usr/bin/env python3
"""HBM traffic per launch of one kernel from two rocprofv3 PMC passes of bench.py (FETCH_SIZE and WRITE_SIZE cannot
share a pass on gfx950: MI355X_MICROARCH.md, rocprofv3 PMC slots).
  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of wide coalesced reads
  (16 B / lane, which is what every kernel here issues), so the read side is doubled; WRITE_SIZE is exact.
usage: summarize_pmc_traffic.py <fetch_dir> <write_dir> <kernel-substring> <launches_in_timed_region> <out_json>"""
import csv
import glob
import json
import os
import sys


def per_dispatch(d, counter, sub):
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    agg = {}
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"] and r["Counter_Name"] == counter:
            k = int(r["Dispatch_Id"])
            agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
    return [agg[k] for k in sorted(agg)]


def main():
    fd, wd, sub, n, out = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    fetch = per_dispatch(fd, "FETCH_SIZE", sub)[-n:]
    write = per_dispatch(wd, "WRITE_SIZE", sub)[-n:]
    fb = 2.0 * 1024.0 * sum(fetch) / len(fetch)
    wb = 1024.0 * sum(write) / len(write)
    res = {"kernel": sub, "launches_averaged": len(fetch), "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
           "hbm_bytes_per_launch": round(fb + wb),
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes of bench.py); FETCH_SIZE x2 (gfx950 wide-read correction), KiB -> bytes"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
