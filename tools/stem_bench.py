import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops
torch.manual_seed(0)
x4 = torch.randn(32, 480, 640, 4, device="cuda"); x4[..., 3] = 0
w = torch.randn(64, 3, 7, 7) / 12
bn = (torch.ones(64), torch.zeros(64), torch.zeros(64), torch.ones(64), 1e-5)
p = ops.pack_stem(w, bn)
ref = None
for tune in (3, 0, 3, 0):
    y = ops.conv2d(x4, p, tune=tune); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.conv2d(x4, p, tune=tune, out=y)
    e1.record(); torch.cuda.synchronize()
    if ref is None: ref = y.clone()
    print("tune", tune, e0.elapsed_time(e1) / 5, "ms  equal:", torch.equal(ref, y))
