"""ROIAlign backward of the training step (16 images x 512 sampled ROIs, 7x7, 256 channels): tile-gather form vs float atomics."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import train_ops as T  # noqa: E402

B, R, C = (int(sys.argv[1]) if len(sys.argv) > 1 else 16), 512, 256
g = torch.Generator().manual_seed(5)
# proposals as the sampler leaves them: clustered around a few objects per image, sizes 20 .. 400 px
ctr = torch.rand(B, 6, 2, generator=g) * torch.tensor([560.0, 400.0]) + 40
pick = torch.randint(0, 6, (B, R), generator=g)
c = torch.gather(ctr, 1, pick[:, :, None].expand(B, R, 2)) + torch.randn(B, R, 2, generator=g) * 25
wh = torch.rand(B, R, 2, generator=g) * 380 + 20
boxes = torch.cat([c - wh / 2, c + wh / 2], 2).clamp(min=0)
boxes[..., 2].clamp_(max=640)
boxes[..., 3].clamp_(max=480)
boxes = boxes.cuda().contiguous()
dout = torch.randn(B * R, 7, 7, C, device="cuda")
dfe = [torch.zeros(B, 480 // s, 640 // s, C, device="cuda") for s in (4, 8, 16, 32)]
scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
for name, kw in (("gather", {}), ("atomics", dict(scatter=True))):
    ts = []
    for _ in range(7):
        for d in dfe:
            d.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        T.roi_align_fpn_backward(dfe, scales, boxes, dout, P=7, sampling_ratio=0, aligned=True, **kw)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{name}: {sorted(ts)[3]:.3f} ms; sum {sum(float(d.double().sum()) for d in dfe):.6e}", flush=True)
