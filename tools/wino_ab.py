"""A/B of the Winograd GEMM variants (a3d_conv_desc.tune): 201 ping-pong accumulators, 203 two staging sets, 205 one set."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops, _lib
import ctypes as C

def bench(B, H, W, Cin, Cout, tunes=(201, 203, 205)):
    torch.manual_seed(0)
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3) / (9 * Cin) ** 0.5
    p = ops.pack_conv(w, torch.randn(Cout), stride=1, pad=1, act=ops.ACT_RELU)
    ref = None
    out = {}
    for t in tunes * 2:
        ops.CONV_TIMING = []
        for _ in range(6):
            y = ops.conv2d(x, p, wino=True, tune=t)
        torch.cuda.synchronize()
        tm = [a.elapsed_time(b) for n, fl, a, b, *_ in ops.CONV_TIMING if n.startswith("wino_gemm")][1:]
        ops.CONV_TIMING = None
        if ref is None: ref = y.clone()
        out.setdefault(t, []).append(min(tm))
        assert torch.equal(ref, y), t
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    print(f"{B}x{H}x{W}x{Cin}->{Cout}: " + "  ".join(f"tune {t}: {min(v):.3f} ms {fl / min(v) / 1e9:.0f} algTF/s" for t, v in out.items()))

bench(32, 120, 160, 256, 256)
bench(32, 60, 80, 128, 128)
bench(32, 30, 40, 256, 256)
bench(124, 14, 14, 256, 256)
bench(64, 60, 80, 256, 256)
