import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_detector
from articulation3d_amd import ops
from articulation3d_amd.utils.synthetic import synthetic_frames
model, cfg = build_detector(0.5, "cuda:0")
x = torch.from_numpy(synthetic_frames(64)).cuda()
for _ in range(3): model.inference_batched(x)
for mode in ("plain", "events", "plain", "events"):
    torch.cuda.synchronize()
    ops.CONV_TIMING = [] if mode == "events" else None
    t0 = time.perf_counter()
    for _ in range(10): model.inference_batched(x)
    torch.cuda.synchronize()
    ops.CONV_TIMING = None
    print(mode, f"{(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
