"""Probe: capture the batched detector in a HIP graph (torch.cuda.CUDAGraph) and compare eager vs replay."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_detector
from articulation3d_amd.utils.synthetic import synthetic_frames

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
model, _ = build_detector(0.5, "cuda:0")
frames = torch.from_numpy(synthetic_frames(B)).cuda()
static_in = frames.clone()
for _ in range(3):
    out = model.inference_batched(static_in)
torch.cuda.synchronize()
ref_rec, ref_cnt = out.records.clone(), out.rec_count.clone()

def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

print("eager ms", timeit(lambda: model.inference_batched(static_in)))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): model.inference_batched(static_in)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    gout = model.inference_batched(static_in)
g.replay(); torch.cuda.synchronize()
print("equal records", torch.equal(gout.records, ref_rec), "counts", torch.equal(gout.rec_count, ref_cnt))
print("graph ms", timeit(lambda: g.replay()))
static_in.copy_(torch.from_numpy(synthetic_frames(B, seed=77)).cuda())
g.replay(); torch.cuda.synchronize()
e = model.inference_batched(static_in); torch.cuda.synchronize()
print("new input equal", torch.equal(gout.records, e.records), torch.equal(gout.rec_count, e.rec_count))
