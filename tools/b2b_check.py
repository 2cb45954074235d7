"""Dev tool (GPU box): the back-to-back pointwise pair (ops.conv2d_b2b, csrc/conv_xs_b2b.hip) against its two launches --
bit equality of both outputs and of their recorded maxima, and time.   python tools/b2b_check.py [frames]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import articulation3d_amd  # noqa: F401,E402
from articulation3d_amd import ops  # noqa: E402


def timeit(fn, n=10):
    for _ in range(3):
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    torch.manual_seed(0)
    for name, H, W, Cin, Cmid, Cout2 in (("res2", 120, 160, 64, 256, 64), ("res3", 60, 80, 128, 512, 128)):
        if (Cin, Cout2) not in ops.B2B_PAIRS:
            continue
        for M_ in (B,):
            x = torch.relu(torch.randn(M_, H, W, Cin, device="cuda")) * torch.exp(0.7 * torch.randn(Cin, device="cuda"))
            res = torch.relu(torch.randn(M_, H, W, Cmid, device="cuda"))
            ones = lambda c: (1.0 + 0.2 * torch.randn(c), 0.1 * torch.randn(c), 0.05 * torch.randn(c), 1.0 + 0.1 * torch.rand(c), 1e-5)
            p1 = ops.pack_conv(torch.randn(Cmid, Cin, 1, 1) / Cin ** 0.5, None, ones(Cmid), 1, 0, ops.ACT_RELU)
            p2 = ops.pack_conv(torch.randn(Cout2, Cmid, 1, 1) / Cmid ** 0.5, None, ones(Cout2), 1, 0, ops.ACT_RELU)
            # the producers of x / res recorded their maxima; here: computed
            for t in (x, res):
                t._a3d_amax = t.abs().flatten(1).amax(1).contiguous()
            y0 = ops.conv2d(x, p1, res=res)
            v1 = ops.last_conv_variant()
            z0 = ops.conv2d(y0, p2, precision=2)
            v2 = ops.last_conv_variant()
            zf = ops.conv2d(y0, p2)  # the same layer in the default arithmetic (what the step ran before round 6)
            v3 = ops.last_conv_variant()
            pair = ops.conv2d_b2b(x, p1, res, p2)
            assert pair is not None, "pair refused"
            y1, z1 = pair
            torch.cuda.synchronize()
            vb = ops.last_conv_variant()
            same_y, same_z = bool(torch.equal(y0, y1)), bool(torch.equal(z0, z1))
            same_ay = bool(torch.equal(ops.amax_of(y0), ops.amax_of(y1)))
            same_az = bool(torch.equal(ops.amax_of(z0), ops.amax_of(z1)))
            dz = float((z1 - zf).abs().max() / zf.abs().max())
            t_a = timeit(lambda: ops.conv2d(x, p1, res=res))
            t_b2 = timeit(lambda: ops.conv2d(y0, p2, precision=2))
            t_b3 = timeit(lambda: ops.conv2d(y0, p2))
            t_f = timeit(lambda: ops.conv2d_b2b(x, p1, res, p2))
            print(f"{name} {M_}x{H}x{W} {Cin}->{Cmid}->{Cout2}: y equal {same_y} (maxima {same_ay}), z equal to the bf16x3 launch {same_z} (maxima {same_az}), "
                  f"z vs the fp16x2 launch {dz:.2e}", flush=True)
            print(f"   {v1}: {t_a:.3f} ms | {v2}: {t_b2:.3f} ms | {v3}: {t_b3:.3f} ms | {vb}: {t_f:.3f} ms   "
                  f"(two launches fp16x2 {t_a + t_b3:.3f}, fused saves {t_a + t_b3 - t_f:.3f} ms)", flush=True)
            if not (same_y and same_z):
                bad = (z0 != z1)
                print("   mismatching z elements:", int(bad.sum()), "of", bad.numel(), " max abs diff", float((z0 - z1).abs().max()),
                      " y mismatches:", int((y0 != y1).sum()), flush=True)


if __name__ == "__main__":
    main()
