"""Time of ops.linear_small at the shapes of the step (mask predictor: rows x 256 -> 1 + sigmoid; head outputs: rows x 1024 -> 3)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from articulation3d_amd import ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for rows, K, N, sig in ((276 * 784, 256, 1, True), (276, 1024, 3, False), (4 * 784, 256, 1, True), (6400 * 784, 256, 1, True)):
    x = torch.randn(rows, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    us = t(lambda: ops.linear_small(x, w, b, norm_n=0, sigmoid=sig))
    print(f"rows {rows:8d} K {K} N {N}: {us:8.1f} us  {rows * K * 4 / us / 1e6:7.2f} TB/s")
