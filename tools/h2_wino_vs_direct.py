"""fp16x2 mode: a 3x3 s1 p1 layer as Winograd F(2x2,3x3) (input transform + wide GEMM) or as a direct convolution?  Whole-layer times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

SHAPES = [(64, 120, 160, 256, 256), (64, 120, 160, 256, 128), (64, 60, 80, 256, 256), (64, 60, 80, 128, 128), (64, 30, 40, 256, 256), (64, 15, 20, 512, 512),
          (64, 15, 20, 256, 256), (64, 8, 10, 256, 256), (276, 14, 14, 256, 256), (1600, 14, 14, 256, 256)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, H, W, Cin, Cout in SHAPES:
    torch.manual_seed(1)
    x = torch.randn(B, H, W, Cin, device="cuda")
    pk = ops.pack_conv(torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5), torch.randn(Cout) * 0.1, None, 1, 1, ops.ACT_RELU)
    res = []
    for kw in (dict(), dict(wino=False)):
        y = ops.conv2d(x, pk, precision=3, **kw)
        v = ops.last_conv_variant()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv2d(x, pk, precision=3, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res.append((v, sorted(ts)[3], y))
    err = float((res[0][2] - res[1][2]).abs().max() / res[1][2].abs().max())
    print(f"{B}x{H}x{W}x{Cin}->{Cout}: winograd [{res[0][0]}] {res[0][1]:.3f} ms | direct [{res[1][0]}] {res[1][1]:.3f} ms | rel diff {err:.1e}", flush=True)
