#!/usr/bin/env python3
"""Dev tool (GPU box): the one-launch Winograd kernel (csrc/conv_wino_fused.hip) against the two-launch form (tune=7) and
float64, with timings per layer shape of the detector.   python tools/wino_fused_check.py [--frames 32]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from articulation3d_amd import ops  # noqa: E402


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    args = ap.parse_args()
    Bf = args.frames
    shapes = [("res2 64->64", Bf, 120, 160, 64, 64), ("res3 128->128", Bf, 60, 80, 128, 128), ("res4 256->256", Bf, 30, 40, 256, 256),
              ("res5 512->512", Bf, 15, 20, 512, 512), ("fpn p2 256->256", Bf, 120, 160, 256, 256), ("fpn p3", Bf, 60, 80, 256, 256),
              ("fpn p5 odd", Bf, 15, 20, 256, 256), ("heads 14x14 x400", 400, 14, 14, 256, 256), ("ragged 7x9 32->36", 3, 7, 9, 32, 36)]
    for name, B, H, W, Cin, Cout in shapes:
        torch.manual_seed(0)
        x = torch.randn(B, H, W, Cin, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5)
        b = torch.randn(Cout)
        pk = ops.pack_conv(w, b, None, 1, 1, ops.ACT_RELU)
        y_f = ops.conv2d(x, pk)
        v_f = ops.last_conv_variant()
        y_2 = ops.conv2d(x, pk, tune=7)
        v_2 = ops.last_conv_variant()
        nb = min(B, 2)
        ref = F.relu(F.conv2d(x[:nb].permute(0, 3, 1, 2).double().cpu(), w.double(), b.double(), padding=1)).permute(0, 2, 3, 1)
        sc = ref.abs().max()
        e_f = float((y_f[:nb, ..., :Cout].double().cpu() - ref).abs().max() / sc)
        e_2 = float((y_2[:nb, ..., :Cout].double().cpu() - ref).abs().max() / sc)
        d = float((y_f - y_2).abs().max() / sc)
        t_f = timeit(lambda: ops.conv2d(x, pk))
        t_2 = timeit(lambda: ops.conv2d(x, pk, tune=7))
        t_d = timeit(lambda: ops.conv2d(x, pk, wino=False))
        gf = 2.0 * B * H * W * Cout * 9 * Cin / 1e9
        print(f"{name:22s} fused {t_f:7.3f} ms ({gf / t_f:6.1f} alg TF/s)  two-launch {t_2:7.3f} ms  direct {t_d:7.3f} ms | err vs f64: fused {e_f:.2e} "
              f"two-launch {e_2:.2e}  fused-vs-two {d:.2e} | {v_f.split(' ')[0]} / {v_2}", flush=True)


if __name__ == "__main__":
    main()
