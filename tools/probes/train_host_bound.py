"""Is the training step host-bound?  Times the host's enqueue loop (no synchronisation) against the wall time to the end of the GPU work."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench import build_detector
from train_bench import synthetic_targets
from articulation3d_amd.training import DetectorTrainer
from articulation3d_amd.utils.synthetic import synthetic_frames
dev = torch.device("cuda:0")
model, _ = build_detector(0.5, dev)
for B in (2, 16):
    tr = DetectorTrainer(model, seed=2020, precision="bf16")
    frames = torch.from_numpy(synthetic_frames(B, seed=2020)).to(dev)
    tg = synthetic_targets(B, 2020)
    gtb, gtc = [t[0] for t in tg], [t[1] for t in tg]
    for _ in range(5): tr.step(frames, gtb, gtc)
    torch.cuda.synchronize()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K): tr.step(frames, gtb, gtc)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B}: host enqueue {1e3*(t1-t0)/K:.3f} ms/step, wall {1e3*(t2-t0)/K:.3f} ms/step, tail after the last enqueue {1e3*(t2-t1):.3f} ms")
