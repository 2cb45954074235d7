"""Where the gradient-exchange segments sit inside a training step: from a rocprofv3 kernel trace of tools/probes/exchange_probe_run.py
(the last steps of the trace run with `grad_overlap="force"` on a one-rank RCCL group).
  python3 tools/trace_exchange.py <rocprof output dir> [steps]
Per step (relative to the step's first kernel): when each segment's cast starts on the communication stream, when the main stream's last
kernel in front of the optimiser ends (= the backward pass is over), when the SGD kernel starts."""
import csv, glob, statistics, sys

d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("preprocess")]
out = []
for a, b in list(zip(idx, idx[1:]))[-n:]:
    step = rows[a:b]
    t0 = int(step[0]["Start_Timestamp"])
    rel = lambda r, k="Start_Timestamp": (int(r[k]) - t0) / 1e6
    sgd = [r for r in step if "sgd_kernel" in r["Kernel_Name"]]
    casts = [r for r in step if "f32_to_bf16_scaled" in r["Kernel_Name"]]
    if not sgd or len(casts) < 4:
        continue
    main = step[0]["Stream_Id"]
    # (the step's own segment casts run in front of its SGD kernel; the casts BEHIND it are the next step's filter copies)
    seg = [r for r in casts if int(r["Start_Timestamp"]) < int(sgd[0]["Start_Timestamp"])][-4:]
    if len(seg) < 4:
        continue
    last_bwd = max((r for r in step if r["Stream_Id"] == main and int(r["Start_Timestamp"]) < int(sgd[0]["Start_Timestamp"]) and "sgd" not in r["Kernel_Name"]),
                   key=lambda r: int(r["End_Timestamp"]))
    out.append(([rel(r) for r in seg], rel(last_bwd, "End_Timestamp"), rel(sgd[0]), rel(sgd[0], "End_Timestamp")))
med = lambda xs: statistics.median(xs)
print(f"{len(out)} steps with the segmented exchange (one-rank RCCL group), times in ms from the step's first kernel, medians:")
for k, name in enumerate(("box head", "res5 + FPN + RPN head", "res4", "res3")):
    print(f"  segment {k} ({name:22s}) cast starts at {med([o[0][k] for o in out]):6.3f}")
print(f"  main stream: last kernel of the backward pass ends at {med([o[1] for o in out]):6.3f}")
print(f"  SGD kernel (reads the bf16 sum) {med([o[2] for o in out]):6.3f} .. {med([o[3] for o in out]):6.3f}")
print(f"  behind the backward pass: {med([o[2] - o[1] for o in out]):6.3f} ms until the optimiser starts (under the profiler)")
