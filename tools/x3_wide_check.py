"""Wide (256 x 256 tiles, pre-split weights by LDS-DMA) vs narrow (128 x 128, split on the fly) split-operand direct kernel:
bit equality and time per layer.  Developer tool; the equality is also a GPU test (tests/test_gpu_parity.py).
shape = B x H x W x Cin x Cout x k x stride"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

PREC = int(os.environ.get("X3_PREC", "3"))  # 3 = fp16x2 (default), 2 = bf16x3

SHAPES = [(64000, 1, 1, 12544, 1024, 1, 1), (64000, 1, 1, 1024, 1024, 1, 1), (64, 30, 40, 256, 1024, 1, 1), (64, 30, 40, 1024, 256, 1, 1),
          (64, 60, 80, 128, 512, 1, 1), (64, 120, 160, 256, 256, 1, 1), (64, 15, 20, 512, 2048, 1, 1), (64, 15, 20, 2048, 512, 1, 1),
          (64, 30, 40, 1024, 2048, 1, 2), (64, 120, 160, 64, 256, 1, 1), (64, 60, 80, 512, 1024, 1, 2)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]


def ms(x, pk, tune, res, reps=7):
    ops.CONV_TIMING = []
    for _ in range(reps):
        ops.conv2d(x, pk, precision=PREC, tune=tune, res=res)
    torch.cuda.synchronize()
    t, ops.CONV_TIMING = ops.CONV_TIMING, None
    g = sorted(a.elapsed_time(b) for (_n, _f, a, b, *_r) in t)
    return g[len(g) // 2]


for B, H, W, Cin, Cout, k, st in SHAPES:
    torch.manual_seed(1)
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, k, k) / (k * Cin ** 0.5)
    pk = ops.pack_conv(w, torch.randn(Cout) * 0.1, None, st, k // 2, ops.ACT_RELU)
    Ho, Wo = (H + 2 * (k // 2) - k) // st + 1, (W + 2 * (k // 2) - k) // st + 1
    res = torch.randn(B, Ho, Wo, pk.cols, device="cuda") if (Cout >= 4 * Cin or Cout == Cin) else None
    y9 = ops.conv2d(x, pk, precision=PREC, tune=9, res=res)
    v9 = ops.last_conv_variant()
    y8 = ops.conv2d(x, pk, precision=PREC, tune=8, res=res)
    v8 = ops.last_conv_variant()
    y0 = ops.conv2d(x, pk, precision=PREC, res=res)
    v0 = ops.last_conv_variant()
    same = torch.equal(y9, y8) and torch.equal(y0, y8)
    m9, m8 = ms(x, pk, 9, res), ms(x, pk, 8, res)
    tf = 2.0 * B * Ho * Wo * pk.cols * Cin * k * k / 1e9
    print(f"{B}x{H}x{W}x{Cin}->{Cout} k{k} s{st}{' +res' if res is not None else ''}: {v9} {m9:.3f} ms ({tf / m9:.0f} TF/s fp32-eq) | {v8} {m8:.3f} ms "
          f"({tf / m8:.0f}) | default {v0} | bit-equal {same}", flush=True)
