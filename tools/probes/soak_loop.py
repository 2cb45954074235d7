"""Soak: the reference's per-frame loop over many frames -- memory stays flat, rate stays flat, results repeat."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench import build_detector
from loop_bench import reference_loop, _Holder
from articulation3d_amd.utils.arti_vis import PlaneRCNN_Branch
from articulation3d_amd.utils.synthetic import synthetic_frames

model, cfg = build_detector(0.5, torch.device("cuda:0"))
branch = PlaneRCNN_Branch(cfg, load_weights=False, predictor=_Holder(model))
frames = synthetic_frames(96, 2020)
first = None
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    torch.cuda.synchronize(); t = time.perf_counter()
    recs = reference_loop(branch, frames)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / len(frames)
    sig = [(len(r.pred_boxes), float(r.pred_boxes.tensor.sum()) if len(r.pred_boxes) else 0.0, float(r.scores.sum()) if len(r.pred_boxes) else 0.0) for r in recs]
    if first is None: first = sig
    print(f"pass {rep}: {1 / dt:6.1f} frames/s  allocated {torch.cuda.memory_allocated() / 2**20:8.1f} MiB  reserved {torch.cuda.memory_reserved() / 2**20:8.1f} MiB  "
          f"same results as pass 0: {sig == first}", flush=True)
