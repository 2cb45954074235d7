"""Per-detection spread between the arithmetic modes on the end-to-end test's frames (threshold 0.0: 100 detections per frame):
how far do the fp16x2 / bf16x3 head outputs sit from the fp32-input MFMA's, detection by detection?"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402
from bench import build_detector  # noqa: E402
from oracle import planercnn_oracle as O  # noqa: E402  (developer tool: synthetic frames only)

seeds = [int(f["seed"] if isinstance(f, dict) else f) for f in json.load(open("tests/golden/e2e_frames.json"))["frames"]][:12]
model, cfg = build_detector(0.0, "cuda:0")
P = O.init_params(2020)
model.load_state_dict(P, strict=False)
res = {}
for mode in (0, 2, 3):
    ops.DEFAULT_PRECISION = mode
    outs = []
    for s in seeds:
        fr = torch.from_numpy(O.synthetic_frames(1, seed=s)).cuda()
        o = model.inference_batched(fr)
        outs.append((o.det.boxes[0].cpu(), o.det.pred_tran_axis.cpu(), o.det.pred_rot_axis.cpu(), o.det.pred_plane.cpu()))
    res[mode] = outs
for name, idx in (("tran_axis", 1), ("rot_axis", 2), ("plane", 3)):
    for mode in (2, 3):
        devs = []
        for a, b in zip(res[mode], res[0]):
            if a[0].shape == b[0].shape and torch.equal((a[0] - b[0]).abs() < 0.05, torch.ones_like(a[0], dtype=torch.bool)):
                devs.append((a[idx] - b[idx]).abs().amax(1))
        d = torch.cat(devs)
        q = torch.quantile(d, torch.tensor([0.5, 0.9, 0.99]))
        print(f"{name:10s} mode {mode} vs fp32 MFMA: n={d.numel()} median {q[0]:.2e} p90 {q[1]:.2e} p99 {q[2]:.2e} max {d.max():.2e}")
