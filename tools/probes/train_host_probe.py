"""2-image training step: host enqueue time against the synchronised step, and where the host time goes (cProfile)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench import build_detector
from train_bench import synthetic_targets
from articulation3d_amd.training import DetectorTrainer
from articulation3d_amd.utils.synthetic import synthetic_frames

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
model, _ = build_detector(0.5, dev)
tr = DetectorTrainer(model, seed=2020, precision="bf16")
frames = torch.from_numpy(synthetic_frames(B, seed=2020)).to(dev)
tg = synthetic_targets(B, 2020); gtb, gtc = [t[0] for t in tg], [t[1] for t in tg]
for _ in range(5): tr.step(frames, gtb, gtc)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N): tr.step(frames, gtb, gtc)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"B={B}: enqueue per step {(t1 - t0) / N * 1e3:.2f} ms; with final sync {(t2 - t0) / N * 1e3:.2f} ms")
tot = 0.0
for _ in range(N):
    a = time.perf_counter(); tr.step(frames, gtb, gtc); torch.cuda.synchronize(); tot += time.perf_counter() - a
print(f"synchronised per step {tot / N * 1e3:.2f} ms")
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for _ in range(N): tr.step(frames, gtb, gtc)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue()[:9000])
