"""fp16x2 mode: a 3x3 s1 p1 layer as Winograd F(2x2,3x3) (input transform + wide GEMM) or as a direct convolution?  Whole-layer times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

SHAPES = [(64, 120, 160, 256, 256), (64, 120, 160, 256, 128), (64, 60, 80, 256, 256), (64, 60, 80, 256, 128), (64, 60, 80, 128, 128), (64, 30, 40, 256, 256),
          (64, 30, 40, 256, 128), (64, 15, 20, 512, 512), (64, 15, 20, 256, 256), (64, 15, 20, 256, 128), (64, 8, 10, 256, 256), (276, 14, 14, 256, 256),
          (1600, 14, 14, 256, 256), (6400, 14, 14, 256, 256)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, H, W, Cin, Cout in SHAPES:
    torch.manual_seed(1)
    x = torch.randn(B, H, W, Cin, device="cuda")
    pk = ops.pack_conv(torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5), torch.randn(Cout) * 0.1, None, 1, 1, ops.ACT_RELU)
    # (the two forms are timed in turn, one call each per round: timed one after the other, the second reads 10-15 % slow)
    kws = (dict(wino=True), dict(wino=False), dict(wino=False, tune=9))  # (tune 9: the wide 256 x 256 direct kernel, whatever K)
    res, ts = [], [[] for _ in kws]
    for kw in kws:
        y = ops.conv2d(x, pk, precision=3, **kw)
        res.append([ops.last_conv_variant(), None, y])
    for _ in range(9):
        for i, kw in enumerate(kws):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv2d(x, pk, precision=3, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts[i].append(e0.elapsed_time(e1))
    for i in range(len(kws)):
        res[i][1] = sorted(ts[i])[4]
    err = float((res[0][2] - res[1][2]).abs().max() / res[1][2].abs().max())
    same = bool(torch.equal(res[1][2], res[2][2]))
    print(f"{B}x{H}x{W}x{Cin}->{Cout}: winograd [{res[0][0]}] {res[0][1]:.3f} ms | direct [{res[1][0]}] {res[1][1]:.3f} ms | wide direct [{res[2][0]}] "
          f"{res[2][1]:.3f} ms (bits equal to narrow: {same}) | rel diff {err:.1e}", flush=True)
