"""Wide (64 tiles x 128 channels, DMA-staged weights) vs 64-wide split-operand Winograd GEMM: bit equality and time per layer.
Developer tool; the equality is also a GPU test (tests/test_gpu_parity.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

PREC = int(os.environ.get("X3_PREC", "3"))  # 3 = fp16x2 (default), 2 = bf16x3

SHAPES = [(64, 120, 160, 256, 256), (64, 60, 80, 256, 256), (64, 60, 80, 128, 128), (64, 30, 40, 256, 256), (64, 15, 20, 512, 512),
          (64, 120, 160, 64, 64), (1600, 14, 14, 256, 256), (64, 120, 160, 256, 128)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]


def gemm_ms(x, pk, tune, reps=10):
    ops.CONV_TIMING = []
    for _ in range(reps):
        ops.conv2d(x, pk, precision=PREC, tune=tune)
    torch.cuda.synchronize()
    t, ops.CONV_TIMING = ops.CONV_TIMING, None
    g = [a.elapsed_time(b) for (name, _, a, b, *_r) in t if name.startswith("wino_gemm")]
    return sorted(g)[len(g) // 2], t[-1][0]


for B, H, W, Cin, Cout in SHAPES:
    torch.manual_seed(1)
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5)
    pk = ops.pack_conv(w, torch.randn(Cout) * 0.1, None, 1, 1, ops.ACT_RELU)
    y0 = ops.conv2d(x, pk, precision=PREC)
    v0 = ops.last_conv_variant()
    y8 = ops.conv2d(x, pk, precision=PREC, tune=8)
    v8 = ops.last_conv_variant()
    same = torch.equal(y0, y8)
    m0, _ = gemm_ms(x, pk, 0)
    m8, _ = gemm_ms(x, pk, 8)
    tiles = B * ((H + 1) // 2) * ((W + 1) // 2)
    tf = 2.0 * tiles * 16 * pk.cols * Cin / 1e9
    print(f"{B}x{H}x{W}x{Cin}->{Cout}: {v0} {m0:.3f} ms ({tf / m0:.0f} TF/s fp32-eq) | {v8} {m8:.3f} ms ({tf / m8:.0f}) | bit-equal {same}", flush=True)
