import os
import sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from articulation3d_amd import ops
torch.manual_seed(0)
K, N, M = 50176, 1024, 6400
pk = ops.pack_linear(torch.randn(N, K) / K ** 0.5, torch.randn(N) * 0.1, act=ops.ACT_RELU)
x = torch.randn(M, 1, 1, K, device="cuda")
ops.amax_of(x)
for _ in range(4):
    ops.conv2d(x, pk, splitk=32)
torch.cuda.synchronize()
