"""Can one batch-1 detector step be captured into a HIP graph (torch.cuda.CUDAGraph) and replayed?  Time eager vs replay, compare outputs."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_detector
from articulation3d_amd.utils.synthetic import synthetic_frames

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
model, cfg = build_detector(0.5, "cuda:0")
x = torch.from_numpy(synthetic_frames(B)).cuda()
for _ in range(3):
    out = model.inference_batched(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    out = model.inference_batched(x)
torch.cuda.synchronize()
print(f"eager B={B}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per step")
ref = (out.rec_count.clone(), out.det.boxes.clone(), out.depth.clone())
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out_g = model.inference_batched(x)
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:300])
    sys.exit(0)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f"graph replay B={B}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per step")
print("same counts:", torch.equal(out_g.rec_count, ref[0]), "same boxes:", torch.equal(out_g.det.boxes, ref[1]), "same depth:", torch.equal(out_g.depth, ref[2]))
