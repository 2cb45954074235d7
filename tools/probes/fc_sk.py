import os
import sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from articulation3d_amd import ops
torch.manual_seed(0)
K, N = 50176, 1024
w = torch.randn(N, K) / K ** 0.5
pk = ops.pack_linear(w, torch.randn(N) * 0.1, act=ops.ACT_RELU)
for M in (276, 6400):
    x = torch.randn(M, K, device="cuda")
    for sk in (16, 32, 64, 128):
        y = ops.linear(x, pk, splitk=sk); v = ops.last_conv_variant()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.linear(x, pk, splitk=sk); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        print(f"M={M} splitk={sk}: {sorted(ts)[3]:.3f} ms [{v}]")
