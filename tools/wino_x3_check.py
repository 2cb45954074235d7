"""Split-operand Winograd GEMM (precision 2 on a Winograd layer) against float64, next to the fp32 Winograd and direct forms."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops
torch.manual_seed(0)
for (B,H,W,Cin,Cout) in [(2,120,160,256,256),(3,61,79,64,64),(3,31,41,128,200),(5,14,14,256,256)]:
    x = torch.randn(B,H,W,Cin,device="cuda"); w = torch.randn(Cout,Cin,3,3)/(Cin*9)**0.5; bias = torch.randn(Cout)
    p = ops.pack_conv(w, bias, None, 1, 1, ops.ACT_NONE)
    ref = F.conv2d(x.permute(0,3,1,2).double().cpu(), w.double(), bias.double(), 1, 1).permute(0,2,3,1)
    e = lambda y: ((y.double().cpu()-ref).norm()/ref.norm()).item()
    print((B,H,W,Cin,Cout), "rel L2 vs float64: wino fp32 %.3e  wino x3 %.3e  direct fp32 %.3e" % (e(ops.conv2d(x,p,precision=0)), e(ops.conv2d(x,p,precision=2)), e(ops.conv2d(x,p,precision=0,wino=False))))
