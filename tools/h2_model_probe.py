import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ["A3D_PRECISION"] = "3"
from articulation3d_amd import ops
from bench import build_detector
from articulation3d_amd.utils.synthetic import synthetic_frames
model, cfg = build_detector(0.5, "cuda:0")
frames = torch.from_numpy(synthetic_frames(8)).cuda()
model.inference_batched(frames)
ops.AMAX_MISSES = []
ops.CONV_TIMING = []
o = model.inference_batched(frames)
torch.cuda.synchronize()
t, ops.CONV_TIMING = ops.CONV_TIMING, None
print("misses:", ops.AMAX_MISSES)
from collections import Counter
print(Counter(n for (n, *_r) in t))
print("det counts", o.det.count.tolist(), "rec", o.rec_count.tolist())
ops.DEFAULT_PRECISION = 2
o2 = model.inference_batched(frames)
print("bf16x3 det counts", o2.det.count.tolist(), "rec", o2.rec_count.tolist())
print("max depth diff", float((o.depth - o2.depth).abs().max()), "records diff", float((o.records - o2.records).abs().max()) if o.records.shape == o2.records.shape else "shape")
