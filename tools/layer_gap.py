"""Every conv / linear launch of one detector batch against ITS OWN floor: max(algorithmic HBM bytes / 8 TB/s, executed matrix FLOPs /
2.5 PFLOP/s).  Sorted by the time above the floor -- where the step loses its milliseconds.  (HIP events per launch: kernels run alone.)
    python tools/layer_gap.py [frames] [threshold]"""
import os
import re
import sys

import torch

os.environ.setdefault("A3D_DEPTH_OVERLAP", "0")
os.environ.setdefault("A3D_HEADS_CONCURRENT_ROWS", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402
from bench import build_detector  # noqa: E402
from articulation3d_amd.utils.synthetic import synthetic_frames  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
model, cfg = build_detector(thr, "cuda:0")
frames = torch.from_numpy(synthetic_frames(B)).cuda()
for _ in range(2):
    model.inference_batched(frames)
torch.cuda.synchronize()
runs = []
for _ in range(3):
    ops.CONV_TIMING = []
    model.inference_batched(frames)
    torch.cuda.synchronize()
    runs.append([(n, a.elapsed_time(b), shape, ex, pipe) for n, fl, a, b, shape, ex, pipe, _st in ops.CONV_TIMING])
ops.CONV_TIMING = None
rows = {}
for i, (name, _, shape, ex, pipe) in enumerate(runs[0]):
    ms = sorted(r[i][1] for r in runs)[1]
    m = re.match(r"(\d+)x(\d+)x(\d+)x(\d+)->(\d+) k(\d+) s(\d+)( ups)?", shape)
    b, h, w, cin, cout, k, s = (int(v) for v in m.groups()[:7])
    ups = bool(m.group(8))
    ho, wo = (2 * h, 2 * w) if ups else ((h + s - 1) // s, (w + s - 1) // s)
    if name.startswith("wino_input"):
        byt = 4.0 * b * h * w * cin + 4.0 * b * ((h + 1) // 2) * ((w + 1) // 2) * 16 * cin
        fl = 0.0
    else:
        byt = 4.0 * b * h * w * cin + 4.0 * b * ho * wo * cout + 4.0 * cout * k * k * cin
        if "wino_gemm" in name:
            byt = 4.0 * b * ((h + 1) // 2) * ((w + 1) // 2) * 16 * cin + 4.0 * b * ho * wo * cout + 4.0 * 16 * cout * cin
        fl = ex * {"f16x3": 3, "bf16x6": 6}.get(pipe, 1)
    floor = max(byt / 8e12, fl / 2.5e15) * 1e3
    key = (name[:44], shape)
    r = rows.setdefault(key, [0, 0.0, 0.0, byt, fl])
    r[0] += 1
    r[1] += ms
    r[2] += floor
print(f"{'kernel':44s} {'shape':40s} {'n':>3s} {'ms':>7s} {'floor':>7s} {'above':>7s} {'x':>5s} {'TB/s':>5s} {'PF/s':>5s}")
tot = [0.0, 0.0]
for (name, shape), (n, ms, floor, byt, fl) in sorted(rows.items(), key=lambda kv: kv[1][2] - kv[1][1]):
    tot[0] += ms
    tot[1] += floor
    print(f"{name:44s} {shape:40s} {n:3d} {ms:7.3f} {floor:7.3f} {ms - floor:7.3f} {ms / max(floor, 1e-9):5.1f} "
          f"{byt * n / ms / 1e9:5.2f} {fl * n / ms / 1e12:5.2f}")
print(f"total {tot[0]:.2f} ms against floors {tot[1]:.2f} ms")
