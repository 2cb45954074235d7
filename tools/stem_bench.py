"""The ResNet stem at 64 frames: conv launch + pool launch vs the one-launch form (a3d_stem_conv_pool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
x4 = torch.randn(B, 480, 640, 4, device="cuda")
x4[..., 3] = 0
w = torch.randn(64, 3, 7, 7) / 12
bn = (torch.ones(64), torch.zeros(64), torch.zeros(64), torch.ones(64), 1e-5)
p = ops.pack_stem(w, bn)
forms = [("conv + pool", lambda: ops.maxpool3x3s2(ops.conv2d(x4, p))), ("one launch", lambda: ops.stem_pool(x4, p))]
outs = [f() for _n, f in forms]
ts = [[] for _ in forms]
for _ in range(7):
    for i, (_n, f) in enumerate(forms):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        ts[i].append(e0.elapsed_time(e1))
print(f"{B} frames: " + " | ".join(f"{n} {sorted(t)[3]:.3f} ms" for (n, _f), t in zip(forms, ts)) + f" | equal bits: {bool(torch.equal(outs[0], outs[1]))}", flush=True)
