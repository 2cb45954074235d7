"""Pointwise launches of the training step's bf16 arithmetic at 16 images per GPU (B=2: at 2): conv_bf16_kernel / conv_bf16w_kernel (tune 34: what
ran before) | conv_bf16xs_kernel (tune 33: activations stationary in registers, csrc/conv_bf16xs.hip) | what the launcher picks (tune 0).
Bits compared, launches timed with HIP events (median of 9, interleaved); the last column is the HBM rate of the fastest form over the
layer's algorithmic bytes (input + output + residual + gate, once each)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

B = int(os.environ.get("B", "16"))
bf, f32 = torch.bfloat16, torch.float32
#        name                                   H    W    Cin  Cout  x    y   res    gate   act
LAYERS = [("res3 conv3 128->512 +res", 60, 80, 128, 512, bf, bf, True, False, True),
          ("res3 conv1 dgrad 128->512 +res gate", 60, 80, 128, 512, bf, bf, True, True, False),
          ("res4 conv3 256->1024 +res", 30, 40, 256, 1024, bf, bf, True, False, True),
          ("res4 conv1 dgrad 256->1024 +res gate", 30, 40, 256, 1024, bf, bf, True, True, False),
          ("res4 conv1 dgrad 256->1024 gate (fp32 in)", 30, 40, 256, 1024, f32, bf, False, True, False),
          ("lateral2 dgrad 256->256 +res (fp32 in)", 120, 160, 256, 256, f32, bf, True, False, False),
          ("res3 shortcut dgrad 256->512 gate (fp32 in)", 60, 80, 256, 512, f32, bf, False, True, False),
          ("rpn predictors dgrad 32->256 gate (fp32 in)", 120, 160, 32, 256, f32, bf, False, True, False),
          ("rpn predictors dgrad 32->256 gate p3", 60, 80, 32, 256, f32, bf, False, True, False),
          ("res3 conv1 512->128", 60, 80, 512, 128, bf, bf, False, False, True),
          ("res3 conv3 dgrad 512->128 gate", 60, 80, 512, 128, bf, bf, False, True, False),
          ("lateral3 512->256", 60, 80, 512, 256, bf, f32, False, False, False),
          ("res5 conv3 512->2048 +res", 15, 20, 512, 2048, bf, bf, True, False, True),
          ("res4 conv1 dgrad 256->1024 +res gate", 30, 40, 256, 1024, bf, bf, True, True, False)]
TUNES = [int(t) for t in os.environ.get("TUNES", "34,33,0").split(",")]
tot = {t: 0.0 for t in TUNES}
ops.BF16_SPLITK_AUTO = True
for name, H, W, Cin, Cout, xdt, odt, has_res, has_gate, relu in LAYERS:
    torch.manual_seed(1)
    x = torch.randn(B, H, W, Cin, device="cuda").to(xdt)
    pk = ops.pack_conv(torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5, torch.randn(Cout) * 0.1, None, 1, 0, ops.ACT_RELU if relu else ops.ACT_NONE)
    pk.w_b16 = pk.w.to(torch.bfloat16)
    res = torch.randn(B, H, W, Cout, device="cuda").to(bf) if has_res else None
    gate = torch.randn(B, H, W, Cout, device="cuda").to(bf) if has_gate else None
    kw = dict(precision=1, res=res, gate=gate, out_dtype=odt)
    outs, labels, times = {}, {}, {t: [] for t in TUNES}
    for t in TUNES:
        outs[t] = ops.conv2d(x, pk, tune=t, **kw).float().clone()
        labels[t] = ops.last_conv_variant()
    for _ in range(9):
        for t in TUNES:
            ops.CONV_TIMING = []
            ops.conv2d(x, pk, tune=t, **kw)
            torch.cuda.synchronize()
            times[t].append(sum(e[2].elapsed_time(e[3]) for e in ops.CONV_TIMING))
    ops.CONV_TIMING = None
    ref = outs[TUNES[0]]
    M = B * H * W
    nbytes = M * (Cin * x.element_size() + Cout * (2 if odt == bf else 4) + (Cout * 2 if has_res else 0) + (Cout * 2 if has_gate else 0))
    cells, best = [], 1e9
    for t in TUNES:
        ms = sorted(times[t])[4]
        tot[t] += ms
        best = min(best, ms)
        cells.append(f"{labels[t]} {ms:.3f} ms" + ("" if torch.equal(outs[t], ref) else f" BITS DIFFER {float((outs[t] - ref).abs().max()):.3g}"))
    print(f"{name:44s} " + " | ".join(cells) + f" | {nbytes / best / 1e9:.2f} TB/s", flush=True)
print("sum: " + " | ".join(f"tune {t}: {tot[t]:.3f} ms" for t in TUNES))
