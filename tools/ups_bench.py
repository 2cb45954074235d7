"""Depth decoder's upsampled 3x3 convs: four phase launches vs the one-launch form (a3d_conv_desc.phase == 5).  Whole-layer times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

# (B, H, W, C1, C2, Cout) of the decoder's five stages at 64 frames (depth_head.py:40-46,72-89)
SHAPES = [(64, 8, 10, 128, 0, 128), (64, 15, 20, 128, 128, 128), (64, 30, 40, 128, 128, 128), (64, 60, 80, 128, 128, 128), (64, 120, 160, 128, 128, 64)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, H, W, C1, C2, Cout in SHAPES:
    torch.manual_seed(1)
    a = torch.randn(B, H, W, C1, device="cuda")
    c2 = torch.randn(B, H, W, C2, device="cuda") if C2 else None
    phases = ops.pack_conv_ups_phases(torch.randn(Cout, C1 + C2, 3, 3) / (3 * (C1 + C2) ** 0.5), torch.randn(Cout) * 0.1, None, ops.ACT_LEAKY)
    forms = [("four launches", dict(fused=False)), ("one launch, tap-outer", dict(fused=True, tune=15)), ("one launch, patch-resident", dict(fused=True, tune=16))]
    outs, ts, names = [], [[] for _ in forms], []
    for _n, kw in forms:
        outs.append(ops.conv2d_ups(a, phases, x2=c2, **kw))
        names.append(ops.last_conv_variant())
    for _ in range(9):
        for i, (_n, kw) in enumerate(forms):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv2d_ups(a, phases, x2=c2, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts[i].append(e0.elapsed_time(e1))
    t4, t1, tp = (sorted(t)[4] for t in ts)
    gf = 2.0 * B * H * W * 4 * Cout * 4 * (C1 + C2) / 1e9
    dflt = ops.conv2d_ups(a, phases, x2=c2)
    vd = ops.last_conv_variant()
    err = float((outs[2] - outs[1]).abs().max() / outs[1].abs().max())
    print(f"{B}x{H}x{W}x({C1}+{C2})->{Cout}: four launches {t4:.3f} ms | tap-outer {t1:.3f} ms | patch-resident {tp:.3f} ms ({gf / tp:.0f} fp32-eq GFLOP/ms) | "
          f"four == tap-outer bits: {bool(torch.equal(outs[0], outs[1]))}; patch-resident vs tap-outer max rel {err:.1e}; default: {vd}", flush=True)
