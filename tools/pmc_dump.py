"""Sums rocprofv3 --pmc counter CSVs per kernel (last dispatch of each kernel name).  python3 tools/pmc_dump.py <dir> [substr] [dispatch index]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
idx = int(sys.argv[3]) if len(sys.argv) > 3 else -1
acc = collections.defaultdict(dict)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if sub in k:
            acc[k].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for c, vals in sorted(v.items()):
        if idx == -2:
            print(f"   {c:32s} " + " ".join(f"{v:.4g}" for v in vals))
        else:
            print(f"   {c:32s} last={vals[idx]:.4g}  n={len(vals)}")
