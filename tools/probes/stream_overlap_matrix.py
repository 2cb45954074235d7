"""Which pairs of a process's HIP streams actually run concurrently?  One spin kernel (a single workgroup, ~2 ms) on each stream of a pair:
the pair takes the time of one if the streams overlap, of two if they are serialised (same hardware queue / pipe).  Streams are torch pool
streams in creation order; 'null' is the default stream.   usage: stream_overlap_matrix.py [number of streams]"""
import sys, time
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
streams = [("null", torch.cuda.default_stream())] + [(f"s{i}", torch.cuda.Stream()) for i in range(n)]
for _, s in streams:  # first use in creation order
    with torch.cuda.stream(s):
        torch.cuda._sleep(1000)
torch.cuda.synchronize()
CY = 4_000_000
def run(ss):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for s in ss:
        with torch.cuda.stream(s):
            torch.cuda._sleep(CY)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t)
one = min(run([streams[0][1]]) for _ in range(3))
print(f"one kernel: {one:.2f} ms")
print("       " + " ".join(f"{nm:>5s}" for nm, _ in streams))
for i, (ni, si) in enumerate(streams):
    row = []
    for j, (nj, sj) in enumerate(streams):
        if j <= i:
            row.append("    .")
        else:
            t = min(run([si, sj]) for _ in range(2))
            row.append(f"{t / one:5.2f}")
    print(f"{ni:>5s}  " + " ".join(row))
