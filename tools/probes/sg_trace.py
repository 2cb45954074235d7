"""Per-launch kernel time of tools/probes/sg_probe.py from a rocprofv3 kernel trace: `sg_trace.py <trace dir> <manifest.json>`."""
import csv, glob, json, statistics, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if any(k in n for k in ("conv_sg_kernel", "conv_x3_kernel", "conv_x3w_kernel", "conv_xs_kernel", "conv_c3p", "conv_xs_b2b", "conv_bf16")):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), n))
rows.sort()
man = json.load(open(sys.argv[2]))
assert sum(c for _, c in man) == len(rows), (sum(c for _, c in man), len(rows))
i = 0
tot = {}
for label, c in man:
    grp = rows[i:i + c]; i += c
    if label == "ref": continue
    per = c // 56 if c % 56 == 0 else 1  # (split-K calls are two launches: the partial sums and their fold)
    calls = [sum(d for _, d, _ in grp[i:i + per]) for i in range(0, len(grp), per)]
    us = statistics.median(calls[6:]) / 1e3
    key = label.split("|")[0].split()[0] + " " + label.split("|")[1].split()[1]
    tot[key] = tot.get(key, 0.) + us
    print(f"{label:90s} {us:7.1f} us")
for k, v in tot.items(): print(f"total {k}: {v:.0f} us")
