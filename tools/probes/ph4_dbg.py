import os
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from articulation3d_amd import ops
for act in (ops.ACT_RELU, ops.ACT_NONE, ops.ACT_LEAKY):
  for (B,H,W,C1,C2,Cout) in [(3,13,42,128,0,64),(3,13,42,128,0,128),(3,13,42,64,0,64),(3,13,42,128,128,64),(2,24,40,128,0,64),(1,3,5,128,0,64)]:
    torch.manual_seed(1)
    spread = torch.logspace(-2, 2, B, device="cuda").view(B,1,1,1)
    x = torch.randn(B,H,W,C1, device="cuda")*spread
    x2 = torch.randn(B,H,W,C2, device="cuda")*spread if C2 else None
    w = torch.randn(Cout, C1+C2, 3,3)/(3*(C1+C2)**0.5)
    ph = ops.pack_conv_ups_phases(w, torch.randn(Cout)*0.1, None, act)
    a = ops.conv2d_ups(x, ph, x2=x2); va = ops.last_conv_variant()
    b = ops.conv2d_ups(x, ph, x2=x2, fused=False); vb = ops.last_conv_variant()
    d = (a != b)
    n = int(d.sum())
    info = ""
    if n:
        idx = d.nonzero()
        info = f" first {idx[0].tolist()} last {idx[-1].tolist()} phases {sorted(set(((i[1]%2)*2+(i[2]%2)).item() for i in idx[:2000]))} imgs {sorted(set(i[0].item() for i in idx[:5000]))} maxabs {float((a-b).abs().max()):.2e} vals {float(a[d][0]):.6g} {float(b[d][0]):.6g}"
    print(f"act {act} {B}x{H}x{W}x({C1}+{C2})->{Cout}: [{va}] vs [{vb}] differing {n}/{a.numel()}{info}")
