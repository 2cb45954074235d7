"""Deep-reduction 1x1 layers of the trunk at 64 frames: conv_h2_kernel (tune 28: the register-staged narrow kernel) | conv_h2dk_kernel with
128-pixel tiles (tune 26) | 256-pixel tiles (tune 27) | what the launcher picks (tune 0).  Bits compared, launch timed with HIP events
(median of 9, interleaved)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

B = int(os.environ.get("B", "64"))
#        name                     H    W    Cin   Cout  stride res
LAYERS = [("res3 conv1 512->128", 60, 80, 512, 128, 1, False),
          ("res3 entry 256->128 s2", 120, 160, 256, 128, 2, False),
          ("res3 shortcut 256->512 s2", 120, 160, 256, 512, 2, False),
          ("res4 conv1 1024->256", 30, 40, 1024, 256, 1, False),
          ("res4 entry 512->256 s2", 60, 80, 512, 256, 2, False),
          ("res4 shortcut 512->1024 s2", 60, 80, 512, 1024, 2, False),
          ("res5 conv1 2048->512", 15, 20, 2048, 512, 1, False),
          ("res5 entry 1024->512 s2", 30, 40, 1024, 512, 2, False),
          ("res5 shortcut 1024->2048 s2", 30, 40, 1024, 2048, 2, False),
          ("res5 conv3 512->2048 +res", 15, 20, 512, 2048, 1, True),
          ("lateral5 2048->256", 15, 20, 2048, 256, 1, False),
          ("lateral4 1024->256", 30, 40, 1024, 256, 1, False),
          ("lateral3 512->256", 60, 80, 512, 256, 1, False)]
TUNES = [int(t) for t in os.environ.get("TUNES", "28,26,27,0").split(",")]
tot = {t: 0.0 for t in TUNES}
for name, H, W, Cin, Cout, s, has_res in LAYERS:
    torch.manual_seed(1)
    x = torch.randn(B, H, W, Cin, device="cuda")
    pk = ops.pack_conv(torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5, torch.randn(Cout) * 0.1, None, s, 0, ops.ACT_RELU)
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    res = torch.randn(B, Ho, Wo, Cout, device="cuda") if has_res else None
    outs, labels = {}, {}
    for t in TUNES:
        outs[t] = ops.conv2d(x, pk, precision=3, tune=t, res=res).clone()
        labels[t] = ops.last_conv_variant()
    times = {t: [] for t in TUNES}
    for _ in range(9):
        for t in TUNES:
            ops.CONV_TIMING = []
            ops.conv2d(x, pk, precision=3, tune=t, res=res)
            torch.cuda.synchronize()
            times[t].append(ops.CONV_TIMING[-1][2].elapsed_time(ops.CONV_TIMING[-1][3]))
    ops.CONV_TIMING = None
    ref = outs[TUNES[0]]
    fl = 2.0 * B * Ho * Wo * Cin * Cout * 3
    cells = []
    for t in TUNES:
        ms = sorted(times[t])[4]
        tot[t] += ms
        cells.append(f"{labels[t]} {ms:.3f} ms ({fl / ms / 1e9:.0f} TF/s)" + ("" if torch.equal(outs[t], ref) else f" BITS DIFFER {float((outs[t] - ref).abs().max()):.3g}"))
    print(f"{name:30s} " + " | ".join(cells), flush=True)
print("sum: " + " | ".join(f"tune {t}: {tot[t]:.3f} ms" for t in TUNES))
