import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from articulation3d_amd import ops
shapes = [(64,30,40,1024,256,1),(64,60,80,512,128,1),(64,15,20,2048,512,1),(64,15,20,512,2048,1),(64,60,80,512,256,1),(64,30,40,1024,2048,2),(64,30,40,256,1024,1)]
for B,H,W,Cin,Cout,st in shapes:
    torch.manual_seed(0)
    x = torch.randn(B,H,W,Cin,device="cuda")
    pk = ops.pack_conv(torch.randn(Cout,Cin,1,1)/Cin**0.5, None, (torch.ones(Cout),torch.zeros(Cout),torch.zeros(Cout),torch.ones(Cout),1e-5), st, 0, ops.ACT_RELU)
    res = []
    for tune in (0, 10, 11, 9):
        try:
            y = ops.conv2d(x, pk, tune=tune, precision=3 if tune else None)
        except RuntimeError as e:
            res.append(f"t{tune}: n/a"); continue
        v = ops.last_conv_variant()
        ts=[]
        for _ in range(7):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record(); ops.conv2d(x, pk, tune=tune, precision=3 if tune else None); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        res.append(f"t{tune} {v.split('_kernel')[0][5:]}{v.split('_kernel')[1]}: {sorted(ts)[3]:.3f}")
    print(f"{B}x{H}x{W}x{Cin}->{Cout} s{st}: " + " | ".join(res), flush=True)
