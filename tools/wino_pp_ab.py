"""A/B of the fp16x2 Winograd GEMM's loop forms on one layer: lockstep (tune 0) | ping-pong (tune 21) | ping-pong + s_setprio (tune 22).
Bits of the three outputs compared, GEMM launch timed with HIP events (median of 9), interleaved so that clock drift hits all alike."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

TUNES = [int(t) for t in os.environ.get("TUNES", "0,21,22").split(",")]
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(64, 120, 160, 256, 256)]
for B, H, W, Cin, Cout in shapes:
    torch.manual_seed(1)
    x = torch.randn(B, H, W, Cin, device="cuda")
    pk = ops.pack_conv(torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5), torch.randn(Cout) * 0.1, None, 1, 1, ops.ACT_RELU)
    PREC = int(os.environ.get("PREC", "3"))
    outs = {t: ops.conv2d(x, pk, precision=PREC, tune=t).clone() for t in TUNES}
    variant = ops.last_conv_variant()
    times = {t: [] for t in TUNES}
    for _ in range(9):
        for t in TUNES:
            ops.CONV_TIMING = []
            ops.conv2d(x, pk, precision=PREC, tune=t)
            torch.cuda.synchronize()
            times[t].append(ops.CONV_TIMING[-1][2].elapsed_time(ops.CONV_TIMING[-1][3]))
    ops.CONV_TIMING = None
    fl = 2.0 * B * ((H + 1) // 2) * ((W + 1) // 2) * 16 * Cout * Cin * (3 if PREC == 3 else 6)
    ref = outs[TUNES[0]]
    print(f"{B}x{H}x{W}x{Cin}->{Cout} [{variant}]: " + " | ".join(
        f"tune {t}: {sorted(times[t])[4]:.3f} ms ({fl / sorted(times[t])[4] / 1e9:.0f} TF/s){'' if torch.equal(outs[t], ref) else ' BITS DIFFER ' + str(float((outs[t] - ref).abs().max()))}"
        for t in TUNES), flush=True)
