"""Dev tool: a few launches of ONE 3x3 layer through the one-launch Winograd kernel (or tune=7: the two-launch form), for
rocprofv3 --pmc passes.   python3 tools/wino_one.py [tune] [frames] [H] [W] [Cin] [Cout]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops

tune = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B, H, W, Cin, Cout = [int(v) for v in sys.argv[2:7]] if len(sys.argv) > 6 else (32, 120, 160, 256, 256)
torch.manual_seed(0)
x = torch.randn(B, H, W, Cin, device="cuda")
w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5)
p = ops.pack_conv(w, torch.randn(Cout), None, 1, 1, ops.ACT_RELU)
for _ in range(4):
    y = ops.conv2d(x, p, tune=tune)
torch.cuda.synchronize()
print(ops.last_conv_variant())
