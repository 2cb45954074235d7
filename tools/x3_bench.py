"""Rate and accuracy of the fp32-grade bf16x3 kernel (a3d_conv_desc.precision == 2) next to the native fp32 kernels.

usage: x3_bench.py [rounds] [filter]
For every shape: median ms / algorithmic TFLOP/s of (a) the default fp32 path (Winograd / pointwise / direct, whatever the
dispatcher picks), (b) the direct fp32 MFMA kernel, (c) bf16x3, (d) plain bf16 (precision 1); and the relative L2 / max
errors of (b), (c), (d) against a float64 evaluation of the same layer on a small batch.
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

SHAPES = [  # name, B, H, W, Cin, Cout, k, stride
    ("fc1 12544->1024", 64000, 1, 1, 12544, 1024, 1, 1),
    ("res2 1x1 64->256", 64, 120, 160, 64, 256, 1, 1),
    ("res2 1x1 256->64", 64, 120, 160, 256, 64, 1, 1),
    ("res3 1x1 128->512", 64, 60, 80, 128, 512, 1, 1),
    ("res4 1x1 1024->256", 64, 30, 40, 1024, 256, 1, 1),
    ("res5 1x1 2048->512", 64, 15, 20, 2048, 512, 1, 1),
    ("p2 3x3 256->256", 64, 120, 160, 256, 256, 3, 1),
    ("res3 3x3 128->128", 64, 60, 80, 128, 128, 3, 1),
    ("res4 3x3 256->256", 64, 30, 40, 256, 256, 3, 1),
    ("res3 3x3 s2 128->128", 64, 120, 160, 128, 128, 3, 2),
]
VARIANTS = [("default", dict()), ("direct", dict(wino=False, tune=5, precision=0)), ("bf16x3", dict(precision=2, wino=False)),
            ("fp16x2", dict(precision=3, wino=False)), ("bf16", dict(precision=1))]
if os.environ.get("X3_BENCH_RELU"):  # realistic activations: post-ReLU with a per-channel spread of magnitudes
    pass


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    torch.manual_seed(0)
    print(f"{'shape':24s} {'GFLOP':>8s} " + " ".join(f"{n:>17s}" for n, _ in VARIANTS) + "   rel-L2 / max-rel error vs float64: direct | bf16x3 | fp16x2 | bf16")
    for name, B, H, W, Cin, Cout, k, s in SHAPES:
        if filt not in name:
            continue
        x = torch.randn(B, H, W, Cin, device="cuda")
        if os.environ.get("X3_BENCH_RELU"):
            x = torch.relu(x) * torch.exp(1.5 * torch.randn(Cin, device="cuda"))
        w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
        bias = torch.randn(Cout)
        p = ops.pack_conv(w, bias, None, s, k // 2, ops.ACT_NONE)
        y = ops.conv2d(x, p)
        fl = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * Cout * Cin * k * k
        times = {n: [] for n, _ in VARIANTS}
        for r in range(rounds + 1):
            for n, kw in VARIANTS:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    ops.conv2d(x, p, out=y, **kw)
                e1.record()
                torch.cuda.synchronize()
                if r:
                    times[n].append(e0.elapsed_time(e1) / 3)
        cells = []
        for n, _ in VARIANTS:
            ms = sorted(times[n])[len(times[n]) // 2]
            cells.append(f"{ms:8.3f}ms {fl / ms / 1e9:6.1f}")
        # accuracy on a slice small enough for a float64 evaluation
        nb = min(B, 2 if H > 1 else 512)
        xs = x[:nb].contiguous()
        cols = F.unfold(xs.permute(0, 3, 1, 2).double(), k, padding=k // 2, stride=s)  # [nb, Cin*k*k, L]
        ref = torch.matmul(w.view(Cout, -1).double().cuda(), cols) + bias.double().cuda()[None, :, None]
        Ho, Wo = y.shape[1], y.shape[2]
        ref = ref.view(nb, Cout, Ho, Wo).permute(0, 2, 3, 1)
        errs = []
        for n, kw in VARIANTS[1:]:
            out = ops.conv2d(xs, p, **kw).double()
            diff = out - ref
            errs.append(f"{(diff.norm() / ref.norm()).item():.2e} / {(diff.abs().max() / ref.abs().max()).item():.2e}")
        print(f"{name:24s} {fl / 1e9:8.1f} " + " ".join(f"{c:>17s}" for c in cells) + "   " + " | ".join(errs), flush=True)


if __name__ == "__main__":
    main()
