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
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    tune = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    torch.manual_seed(0)
    tot_ms = tot_fl = 0.0
    for name, B, H, W, Cin, Cout, k, s, ups in SHAPES:
        if filt not in name:
            continue
        x = torch.randn(B, H, W, Cin, device="cuda")
        w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
        p = ops.pack_conv(w, torch.randn(Cout), None, s, k // 2, ops.ACT_RELU)
        y = ops.conv2d(x, p, ups=bool(ups), tune=tune)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.conv2d(x, p, ups=bool(ups), out=y, tune=tune)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * Cout * Cin * k * k
        tot_ms += ms
        tot_fl += fl
        print(f"{name:28s} {fl / 1e9:9.1f} GFLOP {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF/s", flush=True)
    print(f"{'TOTAL':28s} {tot_fl / 1e9:9.1f} GFLOP {tot_ms:8.3f} ms {tot_fl / tot_ms / 1e9:7.1f} TF/s")


if __name__ == "__main__":
    main()
