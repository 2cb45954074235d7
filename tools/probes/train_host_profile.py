"""cProfile of the host side of the 2-image training step (the step is host-bound there: tools/probes/train_host_bound.py)."""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench import build_detector
from train_bench import synthetic_targets
from articulation3d_amd.training import DetectorTrainer
from articulation3d_amd.utils.synthetic import synthetic_frames
dev = torch.device("cuda:0")
model, _ = build_detector(0.5, dev)
B = int(os.environ.get("B", "2"))
tr = DetectorTrainer(model, seed=2020, precision="bf16")
frames = torch.from_numpy(synthetic_frames(B, seed=2020)).to(dev)
tg = synthetic_targets(B, 2020)
gtb, gtc = [t[0] for t in tg], [t[1] for t in tg]
for _ in range(5): tr.step(frames, gtb, gtc)
torch.cuda.synchronize()
K = 20
pr = cProfile.Profile()
pr.enable()
for _ in range(K): tr.step(frames, gtb, gtc)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(35)
