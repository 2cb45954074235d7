import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from articulation3d_amd import ops
from articulation3d_amd.utils.synthetic import synthetic_frames
for B, H, W in ((64, 480, 640), (5, 96, 128), (7, 100, 132)):
    torch.manual_seed(0)
    fr = torch.from_numpy(synthetic_frames(B)).cuda()[:, :H, :W].contiguous()
    x = ops.preprocess_u8hwc(fr, (103.53, 116.28, 123.675), (57.375, 57.12, 58.395))
    w = torch.randn(64, 3, 7, 7) * 0.05
    bn = (torch.rand(64) + 0.5, torch.randn(64) * 0.1, torch.randn(64) * 0.1, torch.rand(64) + 0.5, 1e-5)
    pk = ops.pack_stem(w, bn)
    out = {}
    for tune in (13, 14):
        y = ops.conv2d(x, pk, precision=3, tune=tune); v = ops.last_conv_variant()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.conv2d(x, pk, precision=3, tune=tune); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        out[tune] = (v, sorted(ts)[3], y, ops.amax_of(y).clone())
    print(f"{B}x{H}x{W}: " + " | ".join(f"{o[0]} {o[1]:.3f} ms" for o in out.values()), "| bits equal:", torch.equal(out[13][2], out[14][2]), "maxima equal:", torch.equal(out[13][3], out[14][3]), flush=True)
