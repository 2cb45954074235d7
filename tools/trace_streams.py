"""Per-stream view of a rocprofv3 kernel trace of tools/train_bench.py: launches, kernel time and idle gaps of every stream over the last N steps
(steps are delimited by the preprocess launch), the union busy time of the GPU, and each stream's largest kernels.
  python3 tools/trace_streams.py <rocprof output dir> [steps]"""
import collections, csv, glob, re, sys

d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 7
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("preprocess")]
sel = rows[idx[-n - 1]:idx[-1]]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(rows[idx[-1]]["Start_Timestamp"])
print(f"{n} steps, {(t1 - t0) / 1e6 / n:.3f} ms per step (under the profiler), {len(sel) / n:.0f} launches per step")
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
busy, (cs, ce) = 0, iv[0]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"GPU busy (union of all streams) {busy / 1e6 / n:.3f} ms per step, idle {(t1 - t0 - busy) / 1e6 / n:.3f}")
name = lambda r: re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "").split("(")[0][:60]
for st in sorted({r["Stream_Id"] for r in sel}):
    rs = [r for r in sel if r["Stream_Id"] == st]
    dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print(f"stream {st}: {len(rs) / n:.0f} launches, kernel time {dur / 1e6 / n:.3f} ms per step")
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rs:
        k = name(r)
        agg[k][0] += 1
        agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"    {k:62s} x{c / n:6.1f} {t / 1e6 / n:7.3f} ms")
