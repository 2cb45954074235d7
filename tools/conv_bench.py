"""Micro-benchmark of the conv-GEMM kernel on representative layer shapes of the detector (random data).
usage: conv_bench.py [reps] [shape-filter-substring]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

SHAPES = [  # name, B, H, W, Cin, Cout, k, stride, ups
    ("p2 3x3 256->256", 32, 120, 160, 256, 256, 3, 1, 0),
    ("res2 3x3 64->64", 32, 120, 160, 64, 64, 3, 1, 0),
    ("res2 1x1 64->256", 32, 120, 160, 64, 256, 1, 1, 0),
    ("res2 1x1 256->64", 32, 120, 160, 256, 64, 1, 1, 0),
    ("res3 3x3 128->128", 32, 60, 80, 128, 128, 3, 1, 0),
    ("res3 1x1 128->512", 32, 60, 80, 128, 512, 1, 1, 0),
    ("res4 3x3 256->256", 32, 30, 40, 256, 256, 3, 1, 0),
    ("res4 1x1 1024->256", 32, 30, 40, 1024, 256, 1, 1, 0),
    ("res5 3x3 512->512", 32, 15, 20, 512, 512, 3, 1, 0),
    ("res5 1x1 2048->512", 32, 15, 20, 2048, 512, 1, 1, 0),
    ("depth deconv5 256->64 ups", 32, 120, 160, 256, 64, 3, 1, 1),
    ("depth conv5 256->128", 32, 120, 160, 256, 128, 3, 1, 0),
    ("fc1 12544->1024", 32000, 1, 1, 12544, 1024, 1, 1, 0),
    ("head conv 3x3 (D=124)", 124, 14, 14, 256, 256, 3, 1, 0),
    ("p6 rpn 3x3", 32, 8, 10, 256, 256, 3, 1, 0),
    ("p5 rpn 3x3", 32, 15, 20, 256, 256, 3, 1, 0),
]


def main():
    """conv_bench.py [rounds] [filter] [tune,tune,...] (tune + 1000 = force the Winograd form): variants are interleaved round by round in ONE process
    (cross-process / cross-device timings are not comparable); prints the median ms and TF/s per variant."""
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    tunes = [int(t) for t in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
    torch.manual_seed(0)
    tot = {t: 0.0 for t in tunes}
    tot_fl = 0.0
    print(f"{'shape':28s} {'GFLOP':>8s} " + " ".join(f"{'t' + str(t):>16s}" for t in tunes))
    for name, B, H, W, Cin, Cout, k, s, ups in SHAPES:
        if filt not in name:
            continue
        x = torch.randn(B, H, W, Cin, device="cuda")
        w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
        p = ops.pack_conv(w, torch.randn(Cout), None, s, k // 2, ops.ACT_RELU)
        y = ops.conv2d(x, p, ups=bool(ups))
        fl = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * Cout * Cin * k * k
        times = {t: [] for t in tunes}
        ok = {t: True for t in tunes}
        for r in range(rounds + 1):
            for t in tunes:
                try:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        ops.conv2d(x, p, ups=bool(ups), out=y, tune=t % 1000, wino=(True if t >= 1000 else None))
                    e1.record()
                    torch.cuda.synchronize()
                    if r:
                        times[t].append(e0.elapsed_time(e1) / 3)
                except RuntimeError:
                    ok[t] = False
        tot_fl += fl
        cells = []
        for t in tunes:
            if ok[t] and times[t]:
                ms = sorted(times[t])[len(times[t]) // 2]
                tot[t] += ms
                cells.append(f"{ms:7.3f}ms {fl / ms / 1e9:6.1f}")
            else:
                cells.append(f"{'n/a':>16s}")
        print(f"{name:28s} {fl / 1e9:8.1f} " + " ".join(f"{c:>16s}" for c in cells), flush=True)
    print(f"{'TOTAL':28s} {tot_fl / 1e9:8.1f} " + " ".join(f"{tot[t]:7.3f}ms {tot_fl / max(tot[t], 1e-9) / 1e9:6.1f}" for t in tunes))


if __name__ == "__main__":
    main()
