"""One fp16x2 Winograd layer: input transform and GEMM timed separately (HIP events), median of 9."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(64, 120, 160, 256, 256)]
for B, H, W, Cin, Cout in shapes:
    torch.manual_seed(1)
    x = torch.randn(B, H, W, Cin, device="cuda")
    pk = ops.pack_conv(torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5), torch.randn(Cout) * 0.1, None, 1, 1, ops.ACT_RELU)
    ops.conv2d(x, pk, precision=3)
    rows = []
    for _ in range(9):
        ops.CONV_TIMING = []
        ops.conv2d(x, pk, precision=3)
        torch.cuda.synchronize()
        rows.append([(t[0], t[2].elapsed_time(t[3])) for t in ops.CONV_TIMING])
    ops.CONV_TIMING = None
    med = lambda i: sorted(r[i][1] for r in rows)[4]
    fl = 2.0 * B * ((H + 1) // 2) * ((W + 1) // 2) * 16 * Cout * Cin * 3
    print(f"{B}x{H}x{W}x{Cin}->{Cout}: " + " | ".join(f"{rows[0][i][0]} {med(i):.3f} ms" for i in range(len(rows[0]))) +
          f" | GEMM {fl / med(len(rows[0]) - 1) / 1e12:.0f} TFLOP/s executed", flush=True)
