"""Small-grid pointwise kernel (csrc/conv_sg_h2.hip, tune 17) against the tiled kernels (tune 10 / 11 = conv_h2_kernel<1> / <2>) and the
dispatcher's pick (tune 0) on the 1x1 layers of the trunk at 1, 2, 4, 8 frames: time per launch and bit equality of outputs and maxima."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from articulation3d_amd import ops


def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3


# name, H, W (input), Cin, Cout, stride, residual
SH = [("res2 conv1 256->64", 120, 160, 256, 64, 1, False), ("res2 conv3 64->256 +res", 120, 160, 64, 256, 1, True),
      ("res3.0 conv1 256->128 s2", 120, 160, 256, 128, 2, False), ("res3 conv1 512->128", 60, 80, 512, 128, 1, False),
      ("res3 conv3 128->512 +res", 60, 80, 128, 512, 1, True), ("res3.0 shortcut 256->512 s2", 120, 160, 256, 512, 2, False),
      ("res4.0 conv1 512->256 s2", 60, 80, 512, 256, 2, False), ("res4 conv1 1024->256", 30, 40, 1024, 256, 1, False),
      ("res4 conv3 256->1024 +res", 30, 40, 256, 1024, 1, True), ("res4.0 shortcut 512->1024 s2", 60, 80, 512, 1024, 2, False),
      ("res5.0 conv1 1024->512 s2", 30, 40, 1024, 512, 2, False), ("res5 conv1 2048->512", 15, 20, 2048, 512, 1, False),
      ("res5 conv3 512->2048 +res", 15, 20, 512, 2048, 1, True), ("res5.0 shortcut 1024->2048 s2", 30, 40, 1024, 2048, 2, False),
      ("lateral5 2048->256", 15, 20, 2048, 256, 1, False), ("lateral4 1024->256", 30, 40, 1024, 256, 1, False),
      ("lateral3 512->256", 60, 80, 512, 256, 1, False), ("lateral2 256->256", 120, 160, 256, 256, 1, False)]
torch.manual_seed(0)
batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 4, 8]
MANIFEST = []  # (label, conv launches) in launch order: tools/probes/sg_trace.py lines a rocprofv3 kernel trace up with it
TUNES = tuple(int(a[6:]) for a in sys.argv[1:] if a.startswith("tunes=")) or (0, 10, 11, 17, 19)
for B in batches:
    tot = {}
    for name, H, W, Cin, Cout, stride, res in SH:
        x = torch.relu(torch.randn(B, H, W, Cin, device="cuda")) * (1 + torch.arange(B, device="cuda").view(B, 1, 1, 1))
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        r = torch.randn(B, Ho, Wo, Cout, device="cuda") if res else None
        p = ops.pack_conv(torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5, torch.randn(Cout) * 0.1, None, stride, 0, ops.ACT_RELU)
        ref = ops.conv2d(x, p, res=r, tune=10 if Cout <= 64 else 11, precision=3); ra = ops.amax_of(ref).clone()
        MANIFEST.append(("ref", 1))
        out = []
        for tune in TUNES:
            try:
                y = ops.conv2d(x, p, res=r, tune=tune, precision=3); v = ops.last_conv_variant()
                eq = bool(torch.equal(y, ref)) and bool(torch.equal(ops.amax_of(y), ra))
                us = t(lambda: ops.conv2d(x, p, res=r, tune=tune, precision=3))
                tot[tune] = tot.get(tune, 0.) + us
                MANIFEST.append((f"B={B} {name} | tune {tune} {v}", 56))
                out.append(f"{tune:2d} {v:22s} {us:6.1f} us {'eq' if eq else 'DIFF'}")
            except Exception as e:
                out.append(f"{tune}: {type(e).__name__} {e}"[:60])
        M = B * Ho * Wo
        print(f"B={B} {name:30s} waves={((M + 31) // 32) * (Cout // 32):6d} " + " | ".join(out), flush=True)
    print(f"B={B} totals (us): " + ", ".join(f"tune {k}: {v:.0f}" for k, v in tot.items()), flush=True)
json.dump(MANIFEST, open(os.environ.get("SG_MANIFEST", "/tmp/sg_manifest.json"), "w"))
