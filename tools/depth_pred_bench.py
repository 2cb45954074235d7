"""deconv5 + depth_pred of the depth head at 64 frames: two layers (patch-resident four-phase conv, then conv3x3_to1) vs the tap-product form
(a3d_conv_desc.dot_w / a3d_tapsum9)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (120, 160)
torch.manual_seed(1)
a = torch.randn(B, H, W, 128, device="cuda")
c2 = torch.randn(B, H, W, 128, device="cuda")
phases = ops.pack_conv_ups_phases(torch.randn(64, 256, 3, 3) / (3 * 256 ** 0.5), torch.randn(64) * 0.1, None, ops.ACT_RELU)
w9 = (torch.randn(3, 3, 64) / 24).cuda()
bias = 0.3
forms = [("two layers", lambda: ops.conv3x3_to1(ops.conv2d_ups(a, phases, x2=c2), w9, bias)), ("tap products", lambda: ops.conv2d_ups_to1(a, phases, w9, bias, x2=c2))]
outs = [f() for _n, f in forms]
ts = [[] for _ in forms]
for _ in range(7):
    for i, (_n, f) in enumerate(forms):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        ts[i].append(e0.elapsed_time(e1))
x64 = ops.conv2d_ups(a, phases, x2=c2).double().permute(0, 3, 1, 2)
ref = torch.nn.functional.conv2d(x64, w9.double().permute(2, 0, 1)[None], torch.tensor([bias], dtype=torch.float64, device="cuda"), 1, 1)[:, 0]
e = [float((o.double() - ref).abs().max() / ref.abs().max()) for o in outs]
print(f"{B}x{H}x{W}: " + " | ".join(f"{n} {sorted(t)[3]:.3f} ms" for (n, _f), t in zip(forms, ts)) + f" | max err vs float64 of the same 64-channel tensor: {e[0]:.1e} | {e[1]:.1e}", flush=True)
