"""fp16x2 direct layers: conv_h2_kernel<2> (128 x 128 tiles, 3 workgroups per CU) or conv_h2w_kernel (256 x 256, tune 9)?  Per-layer A/B
(the two agree bit for bit, so the rule may follow the measurement)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops

# (B, H, W, Cin, Cout, k, stride, residual)
if len(sys.argv) > 1 and sys.argv[1] == 'b32':
    B0 = 32
else:
    B0 = 64
SHAPES = [(64000, 1, 1, 1024, 1024, 1, 1, 0), (64, 15, 20, 2048, 512, 1, 1, 0), (64, 15, 20, 512, 2048, 1, 1, 1), (64, 30, 40, 1024, 256, 1, 1, 0),
          (64, 30, 40, 256, 1024, 1, 1, 1), (64, 60, 80, 512, 128, 1, 1, 0), (64, 60, 80, 128, 512, 1, 1, 1), (64, 120, 160, 256, 256, 1, 1, 0),
          (64, 60, 80, 512, 256, 1, 1, 0), (64, 30, 40, 1024, 256, 1, 1, 0), (64, 15, 20, 2048, 256, 1, 1, 0), (64, 120, 160, 256, 512, 1, 2, 0),
          (64, 60, 80, 512, 1024, 1, 2, 0), (64, 30, 40, 1024, 2048, 1, 2, 0)]
for B, H, W, Cin, Cout, k, st, has_res in SHAPES:
    torch.manual_seed(1)
    x = torch.randn(B, H, W, Cin, device="cuda")
    pk = ops.pack_conv(torch.randn(Cout, Cin, k, k) / (k * Cin ** 0.5), torch.randn(Cout) * 0.1, None, st, k // 2, ops.ACT_RELU)
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    res = torch.randn(B, Ho, Wo, Cout, device="cuda") if has_res else None
    kws = tuple(dict(tune=int(t)) for t in os.environ.get("AB_TUNES", "0,9").split(","))  # 0 dispatcher, 9 wide, 11 narrow 128, 13 narrow 64-pixel
    outs, names, ts = [], [], [[] for _ in kws]
    for kw in kws:
        outs.append(ops.conv2d(x, pk, precision=3, res=res, wino=False, **kw))
        names.append(ops.last_conv_variant())
    for _ in range(9):
        for i, kw in enumerate(kws):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.conv2d(x, pk, precision=3, res=res, wino=False, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts[i].append(e0.elapsed_time(e1))
    tt = [sorted(t)[4] for t in ts]
    print(f"{B}x{H}x{W}x{Cin}->{Cout} k{k} s{st}{' +res' if has_res else ''}: " + " | ".join(f"[{n}] {t:.3f} ms" for n, t in zip(names, tt)) +
          f" | equal bits {all(bool(torch.equal(outs[0], o)) for o in outs[1:])}", flush=True)
