import torch, sys
sys.path.insert(0, '.')
from articulation3d_amd import ops
x = torch.randn(64, 240, 320, 64, device="cuda"); w = torch.randn(9, 64, device="cuda").contiguous()
for _ in range(3): ops.conv3x3_to1(x, w, 0.1)
ts = []
for _ in range(9):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv3x3_to1(x, w, 0.1); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("conv3x3_to1 64x240x320x64:", sorted(ts)[4], "ms")
