"""fp16x2 mode (a3d_conv_desc.precision == 3) checks on a GPU box: error against float64 next to bf16x3, the recorded output maxima,
and batch invariance (an image's result must not depend on what else is in the batch)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["A3D_PRECISION"] = "3"
from articulation3d_amd import ops  # noqa: E402

CASES = [(2000, 1, 1, 4096, 1024, 1, 1), (300, 1, 1, 16384, 1024, 1, 1), (6, 30, 40, 256, 1024, 1, 1), (5, 30, 40, 1024, 256, 1, 1), (4, 60, 80, 128, 128, 3, 2), (300, 1, 1, 1024, 1024, 1, 1), (3, 61, 79, 64, 36, 1, 1),
         (4, 30, 40, 256, 256, 3, 1), (3, 60, 80, 128, 128, 3, 1), (70, 14, 14, 256, 256, 3, 1)]
for B, H, W, Cin, Cout, k, st in CASES:
    torch.manual_seed(2)
    # images of very different magnitude in one batch: the scale is per image
    x = torch.relu(torch.randn(B, H, W, Cin, device="cuda")) * torch.exp(1.5 * torch.randn(Cin, device="cuda")) * torch.logspace(-3, 3, B, device="cuda")[:, None, None, None]
    w = torch.randn(Cout, Cin, k, k) / (k * Cin ** 0.5)
    pk = ops.pack_conv(w, torch.randn(Cout) * 0.1, None, st, k // 2, ops.ACT_RELU)
    sk = ops.choose_splitk(B, pk.cols, Cin) if (H == 1 and k == 1) else 1
    tune = 9 if (Cin >= 4096 and sk == 1) else 0  # (the wide kernel whatever the row count)
    y = ops.conv2d(x, pk, splitk=sk, tune=tune, precision=3 if tune else None)
    v3 = ops.last_conv_variant()
    y2 = ops.conv2d(x, pk, precision=2, splitk=sk, tune=tune)
    ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double().cuda(), pk.shift[:Cout].double(), stride=st, padding=k // 2)).permute(0, 2, 3, 1)
    per = lambda t: ((t[..., :Cout].double() - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)).max().item()  # worst image, relative L2
    amax_ok = torch.equal(y._a3d_amax, y.abs().flatten(1).amax(1))
    sub = x[B // 2:B // 2 + 1].contiguous()
    alone = ops.conv2d(sub, pk, splitk=sk, tune=tune, precision=3 if tune else None)
    inv = torch.equal(alone[0], y[B // 2])
    print(f"{B}x{H}x{W}x{Cin}->{Cout} k{k}s{st}: {v3:28s} rel-L2 worst image fp16x2 {per(y):.2e} | bf16x3 {per(y2):.2e} | y_amax exact {amax_ok} | batch-invariant {inv}", flush=True)
