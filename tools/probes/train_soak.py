"""Soak: N optimiser steps with the side streams against N steps on one stream, same seeds: every parameter must come out bit for bit equal
(a race between the streams would show up as a difference sooner or later)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench import build_detector
from train_bench import synthetic_targets
from articulation3d_amd.training import DetectorTrainer
from articulation3d_amd.utils.synthetic import synthetic_frames
N, B = int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
outs = []
for serial in (True, False):
    torch.manual_seed(2020)
    model, _ = build_detector(0.5, dev)
    tr = DetectorTrainer(model, seed=11, precision="bf16")
    if serial:
        tr._wg_stream = tr._rpn_stream = None
    frames = [torch.from_numpy(synthetic_frames(B, seed=100 + i)).to(dev) for i in range(4)]
    tg = [synthetic_targets(B, 100 + i) for i in range(4)]
    last = None
    for it in range(N):
        k = it % 4
        last, _ = tr.step(frames[k], [t[0] for t in tg[k]], [t[1] for t in tg[k]])
    torch.cuda.synchronize()
    outs.append((tr.params.clone(), {k: float(v) for k, v in last.items()}))
    print("serial" if serial else "streams", outs[-1][1], flush=True)
same = torch.equal(outs[0][0], outs[1][0])
print(f"{N} steps at {B} images: parameters bit-identical = {same}; finite = {bool(torch.isfinite(outs[1][0]).all())}; max |diff| = {float((outs[0][0] - outs[1][0]).abs().max())}")
