"""fp16x2 pointwise layers: the x-stationary kernel (tune 13) against the tiled kernels (tune 14): bits and time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

SHAPES = [(64, 120, 160, 64, 256, 1), (64, 60, 80, 128, 512, 1), (64, 30, 40, 256, 1024, 1), (64, 120, 160, 256, 256, 0), (64, 120, 160, 256, 128, 0),
          (64, 120, 160, 64, 256, 0), (3, 37, 41, 128, 512, 1), (1, 120, 160, 64, 256, 1), (64, 15, 20, 256, 1024, 1)]
for B, H, W, Cin, Cout, with_res in SHAPES:
    torch.manual_seed(1)
    x = torch.randn(B, H, W, Cin, device="cuda") * torch.rand(B, 1, 1, 1, device="cuda") * 3
    res = torch.randn(B, H, W, Cout, device="cuda") if with_res else None
    pk = ops.pack_conv(torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5, torch.randn(Cout) * 0.1, None, 1, 0, ops.ACT_RELU)
    out, ts = {}, {13: [], 14: []}
    for tune in (13, 14):
        out[tune] = ops.conv2d(x, pk, res=res, precision=3, tune=tune)
        out[tune] = (out[tune], ops.last_conv_variant(), ops.amax_of(out[tune]).clone())
    for _ in range(7):
        for tune in (13, 14):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv2d(x, pk, res=res, precision=3, tune=tune)
            e1.record()
            torch.cuda.synchronize()
            ts[tune].append(e0.elapsed_time(e1))
    same = torch.equal(out[13][0], out[14][0]) and torch.equal(out[13][2], out[14][2])
    byt = 4.0 * B * H * W * (Cin + Cout * (2 if with_res else 1))
    t13, t14 = sorted(ts[13])[3], sorted(ts[14])[3]
    print(f"{B}x{H}x{W}x{Cin}->{Cout}{' +res' if with_res else ''}: [{out[13][1]}] {t13:.3f} ms ({byt / t13 / 1e9:.2f} TB/s) | [{out[14][1]}] {t14:.3f} ms "
          f"({byt / t14 / 1e9:.2f} TB/s) | bits equal: {same}", flush=True)
