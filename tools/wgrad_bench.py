"""Weight-gradient launches of the bf16 training step at 16 images per GPU: per-layer times (run once with A3D_WGRAD_TR=0 for the first
form, once without for the transposed-read form of csrc/conv_wgrad_tr.hip) and the error of each against float64 on a small layer."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import train_ops as T  # noqa: E402

# (B, H, W, Cin, Cout, k, x stored bf16, dy stored bf16)
SHAPES = [(16, 120, 160, 256, 256, 3, 0, 0), (16, 60, 80, 256, 256, 3, 0, 0), (16, 30, 40, 256, 256, 3, 0, 0), (16, 15, 20, 256, 256, 3, 0, 0),
          (16, 120, 160, 256, 256, 1, 0, 0), (8192, 1, 1, 12544, 1024, 1, 0, 0), (8192, 1, 1, 1024, 1024, 1, 0, 0), (16, 60, 80, 512, 256, 1, 1, 0),
          (16, 30, 40, 256, 256, 3, 1, 1), (16, 60, 80, 128, 128, 3, 1, 1), (16, 30, 40, 1024, 256, 1, 1, 1), (16, 60, 80, 128, 512, 1, 1, 1),
          (16, 15, 20, 512, 512, 3, 1, 1), (16, 120, 160, 256, 32, 1, 0, 0)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, H, W, Cin, Cout, k, xb, yb in SHAPES:
    torch.manual_seed(3)
    x = torch.randn(B, H, W, Cin, device="cuda")
    dy = torch.randn(B, H, W, Cout, device="cuda")
    if xb:
        x = x.to(torch.bfloat16)
    if yb:
        dy = dy.to(torch.bfloat16)
    dw = torch.empty((Cout, k * k * Cin), device="cuda")
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        T.conv_wgrad(x, dy, dw, KH=k, KW=k, stride=1, pad=k // 2, precision=1)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[3]
    gf = 2.0 * B * H * W * Cout * k * k * Cin / 1e9
    print(f"{B}x{H}x{W}x{Cin}->{Cout} k{k} io{xb + 2 * yb}: {ms:.3f} ms  {gf / ms:.0f} TFLOP/s (launch + slice reduction)", flush=True)
