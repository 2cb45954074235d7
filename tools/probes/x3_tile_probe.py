"""conv_h2_kernel tile choice on the deep 1x1 layers of the 64-frame step: dispatcher's pick (tune 0) against the 128 x 64 (10) and 128 x 128 (11)
tiles, bit equality included."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from articulation3d_amd import ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
SH = [("res3 conv1 512->128", 64, 60, 80, 512, 128, False), ("res4 conv1 1024->256", 64, 30, 40, 1024, 256, False), ("res5 conv1 2048->512", 64, 15, 20, 2048, 512, False),
      ("res5 conv3 512->2048 +res", 64, 15, 20, 512, 2048, True), ("lateral4 1024->256", 64, 30, 40, 1024, 256, False), ("lateral2 256->256", 64, 120, 160, 256, 256, False),
      ("lateral3 512->256", 64, 60, 80, 512, 256, False)]
torch.manual_seed(0)
for name, B, H, W, Cin, Cout, res in SH:
    x = torch.relu(torch.randn(B, H, W, Cin, device="cuda")); r = torch.randn(B, H, W, Cout, device="cuda") if res else None
    p = ops.pack_conv(torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5, torch.randn(Cout) * 0.1, None, 1, 0, ops.ACT_RELU)
    y0 = ops.conv2d(x, p, res=r); v0 = ops.last_conv_variant(); out = []
    for tune in (0, 8, 10, 11, 9):
        try:
            y = ops.conv2d(x, p, res=r, tune=tune, precision=3); v = ops.last_conv_variant()
            out.append(f"tune {tune:2d} {v:22s} {t(lambda: ops.conv2d(x, p, res=r, tune=tune, precision=3)):.3f} ms eq={bool(torch.equal(y, y0))}")
        except Exception as e:
            out.append(f"tune {tune}: {type(e).__name__}")
    print(f"{name:28s} " + " | ".join(out), flush=True)
