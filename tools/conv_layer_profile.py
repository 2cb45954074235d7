"""Per-launch table of the conv-GEMM kernel over one detector batch (HIP events on the launch stream):
shape, algorithmic GFLOP, ms, TFLOP/s.  Developer tool for finding which layers sit below the MFMA roof."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402
from bench import build_detector  # noqa: E402
from articulation3d_amd.utils.synthetic import synthetic_frames  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
model, cfg = build_detector(thr, "cuda:0")
frames = torch.from_numpy(synthetic_frames(B)).cuda()
for _ in range(2):
    model.inference_batched(frames)
torch.cuda.synchronize()
ops.CONV_TIMING = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
model.inference_batched(frames)
e1.record()
torch.cuda.synchronize()
t, ops.CONV_TIMING = ops.CONV_TIMING, None
tot = 0.0
print(f"{'kernel':36s} {'shape':44s} {'GFLOP exec':>10s} {'ms':>8s} {'fp32-eq TF/s':>12s}")
for name, fl, a, b, shape, ex, pipe, _st in t:
    ms = a.elapsed_time(b)
    tot += ms
    print(f"{name[:36]:36s} {shape:44s} {ex / 1e9:10.2f} {ms:8.3f} {ex / ms / 1e9:12.1f}")
print(f"conv total {tot:.2f} ms of step {e0.elapsed_time(e1):.2f} ms")
