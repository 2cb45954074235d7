"""Training-step launches of the bf16 arithmetic at 16 (and 2) images per GPU: conv_bf16_kernel (tune 32) | conv_bf16w_kernel with 128- / 256-channel
tiles (tune 30 / 31) | what the launcher picks (tune 0).  Bits compared, launches timed with HIP events (median of 9, interleaved)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

B = int(os.environ.get("B", "16"))
bf, f32 = torch.bfloat16, torch.float32
#        name                          H    W    Cin   Cout  k  s  x dtype  out dtype  res
LAYERS = [("fpn_output2 3x3 256->256", 120, 160, 256, 256, 3, 1, bf, f32, False),
          ("rpn conv p2 3x3 256->256", 120, 160, 256, 256, 3, 1, f32, bf, False),
          ("fpn_output3 3x3 256->256", 60, 80, 256, 256, 3, 1, bf, f32, False),
          ("res3 conv2 3x3 128->128", 60, 80, 128, 128, 3, 1, bf, bf, False),
          ("res3 conv3 1x1 128->512 +res", 60, 80, 128, 512, 1, 1, bf, bf, True),
          ("res4 conv1 1x1 1024->256", 30, 40, 1024, 256, 1, 1, bf, bf, False),
          ("res4 conv2 3x3 256->256", 30, 40, 256, 256, 3, 1, bf, bf, False),
          ("res4 conv3 1x1 256->1024 +res", 30, 40, 256, 1024, 1, 1, bf, bf, True),
          ("res5 conv2 3x3 512->512", 15, 20, 512, 512, 3, 1, bf, bf, False),
          ("lateral3 1x1 512->256", 60, 80, 512, 256, 1, 1, bf, bf, False),
          ("fc1 12544->1024 (512 rows / image)", 1, 512, 12544, 1024, 1, 1, bf, bf, False),
          ("dgrad fpn_output2 3x3 256->256 fp32 in", 120, 160, 256, 256, 3, 1, f32, bf, False)]
TUNES = [int(t) for t in os.environ.get("TUNES", "32,30,31,0").split(",")]
tot = {t: 0.0 for t in TUNES}
ops.BF16_SPLITK_AUTO = True
for name, H, W, Cin, Cout, k, s, xdt, odt, has_res in LAYERS:
    torch.manual_seed(1)
    x = torch.randn(B, H, W, Cin, device="cuda").to(xdt)
    pk = ops.pack_conv(torch.randn(Cout, Cin, k, k) / (k * Cin ** 0.5), torch.randn(Cout) * 0.1, None, s, k // 2, ops.ACT_RELU)
    pk.w_b16 = pk.w.to(torch.bfloat16)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    res = torch.randn(B, Ho, Wo, Cout, device="cuda").to(odt) if has_res else None
    outs, labels, times = {}, {}, {t: [] for t in TUNES}
    for t in TUNES:
        outs[t] = ops.conv2d(x, pk, precision=1, tune=t, res=res, out_dtype=odt).float().clone()
        labels[t] = ops.last_conv_variant()
    for _ in range(9):
        for t in TUNES:
            ops.CONV_TIMING = []
            ops.conv2d(x, pk, precision=1, tune=t, res=res, out_dtype=odt)
            torch.cuda.synchronize()
            times[t].append(sum(e[2].elapsed_time(e[3]) for e in ops.CONV_TIMING))
    ops.CONV_TIMING = None
    ref = outs[TUNES[0]]
    fl = 2.0 * B * Ho * Wo * Cin * Cout * k * k
    cells = []
    for t in TUNES:
        ms = sorted(times[t])[4]
        tot[t] += ms
        cells.append(f"{labels[t]} {ms:.3f} ms ({fl / ms / 1e9:.0f} TF/s)" + ("" if torch.equal(outs[t], ref) else f" BITS DIFFER {float((outs[t] - ref).abs().max()):.3g}"))
    print(f"{name:40s} " + " | ".join(cells), flush=True)
print("sum: " + " | ".join(f"tune {t}: {tot[t]:.3f} ms" for t in TUNES))
