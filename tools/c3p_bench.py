"""Plain 3x3 s1 p1 fp16x2 layers with fewer than 256 input channels (res2 / res3 conv2): tap-outer kernels (tune 11 / 10) against the
patch-resident kernel (conv_ph4p.hip, PH = false; tune 16 / the dispatcher's choice).  Whole-layer times, agreement."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

SHAPES = [(64, 120, 160, 64, 64), (64, 60, 80, 128, 128), (64, 30, 40, 128, 128), (64, 120, 160, 128, 256)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, H, W, Cin, Cout in SHAPES:
    torch.manual_seed(1)
    x = torch.randn(B, H, W, Cin, device="cuda")
    bn = (torch.rand(Cout) + 0.5, torch.randn(Cout) * 0.1, torch.randn(Cout) * 0.1, torch.rand(Cout) + 0.5, 1e-5)
    pk = ops.pack_conv(torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5), None, bn, 1, 1, ops.ACT_RELU)
    forms = [("tune 11", dict(tune=11)), ("tune 10", dict(tune=10)), ("patch-resident", dict(tune=16)), ("dispatcher", dict())]
    outs, names, ts = [], [], []
    for _n, kw in forms:
        outs.append(ops.conv2d(x, pk, wino=False, precision=3, **kw))
        names.append(ops.last_conv_variant())
        t = []
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv2d(x, pk, wino=False, precision=3, **kw)
            e1.record()
            torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1))
        ts.append(sorted(t)[4])
    err = float((outs[2] - outs[0]).abs().max() / outs[0].abs().max())
    print(f"{B}x{H}x{W}x{Cin}->{Cout}: " + " | ".join(f"{n} [{v}] {t:.3f} ms" for (n, _k), v, t in zip(forms, names, ts)) +
          f" | tune 10 == 11 bits: {bool(torch.equal(outs[0], outs[1]))}; patch-resident vs tap-outer max rel {err:.1e}", flush=True)
