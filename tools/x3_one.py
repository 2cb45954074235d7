import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.manual_seed(0)
x = torch.randn(64000, 1, 1, 12544, device="cuda")
w = torch.randn(1024, 12544, 1, 1) / 112
p = ops.pack_conv(w, torch.randn(1024), None, 1, 0, ops.ACT_NONE)
for _ in range(3):
    y = ops.conv2d(x, p, precision=prec)
torch.cuda.synchronize()
