"""Random-shape fuzz of the round-3 kernels (seeded; a developer tool, the fixed cases live in tests/):
  * conv_h2xs_kernel (tune 13) against the tiled kernels (tune 14): outputs and recorded maxima bit for bit;
  * the fp16x2 Winograd pair (pre-split transform + DMA-ring GEMM) against a float64 convolution: error at the level of the
    direct fp16x2 kernel on the same layer.
    python tools/fuzz_kernels.py [cases] [seed]"""
import os
import random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"  # larger maps / batches: the non-plane-split GEMMs, the N-split tail of conv_h2xs
bad = 0
for case in range(n):
    torch.manual_seed(case)
    B, H, W = (rng.randint(8, 48), rng.randint(40, 130), rng.randint(40, 170)) if BIG else (rng.randint(1, 9), rng.randint(1, 70), rng.randint(1, 90))
    Cin, Cout = rng.choice([64, 128, 256]), rng.choice([128, 256, 384, 512, 1024])
    with_res, act = rng.random() < 0.6, rng.choice([ops.ACT_RELU, ops.ACT_NONE])
    x = torch.randn(B, H, W, Cin, device="cuda") * torch.logspace(-2, 2, B, device="cuda")[:, None, None, None]
    res = torch.randn(B, H, W, Cout, device="cuda") if with_res else None
    pk = ops.pack_conv(torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5, torch.randn(Cout) * 0.1, None, 1, 0, act)
    a = ops.conv2d(x, pk, res=res, precision=3, tune=13)
    va = ops.last_conv_variant()
    b = ops.conv2d(x, pk, res=res, precision=3, tune=14)
    ok = va.startswith("conv_h2xs") and torch.equal(a, b) and torch.equal(ops.amax_of(a), ops.amax_of(b))
    bad += not ok
    print(f"xs   {B}x{H}x{W}x{Cin}->{Cout} res={int(with_res)} act={act}: [{va}] {'ok' if ok else 'MISMATCH'}", flush=True)
for case in range(n):
    torch.manual_seed(1000 + case)
    B, H, W = (rng.randint(4, 24), rng.randint(20, 90), rng.randint(20, 120)) if BIG else (rng.randint(1, 6), rng.randint(2, 50), rng.randint(2, 60))
    Cin, Cout = rng.choice([32, 64, 128, 256]), rng.choice([128, 256, 512])
    x = torch.randn(B, H, W, Cin, device="cuda") * torch.logspace(-1, 1, B, device="cuda")[:, None, None, None]
    w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5)
    bias = torch.randn(Cout) * 0.1
    pk = ops.pack_conv(w, bias, None, 1, 1, ops.ACT_RELU)
    yw = ops.conv2d(x, pk, precision=3, wino=True)
    vw = ops.last_conv_variant()
    yd = ops.conv2d(x, pk, precision=3, wino=False)
    bs = sorted({0, B - 1})  # (float64 reference on the first and last image: the CPU side of the check)
    ref = torch.relu(F.conv2d(x[bs].double().permute(0, 3, 1, 2).cpu(), w.double(), bias.double(), padding=1)).permute(0, 2, 3, 1)
    scale = ref.abs().amax(dim=(1, 2, 3), keepdim=True).clamp_min(1e-30)
    ew = float(((yw[bs].double().cpu() - ref).abs() / scale).max())
    ed = float(((yd[bs].double().cpu() - ref).abs() / scale).max())
    ok = "wino" in vw and ew < max(4 * ed, 3e-6) and bool(torch.isfinite(yw).all())
    bad += not ok
    print(f"wino {B}x{H}x{W}x{Cin}->{Cout}: [{vw}] err {ew:.1e} (direct {ed:.1e}) {'ok' if ok else 'BAD'}", flush=True)
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
