"""bf16 small-grid pointwise kernel (csrc/conv_bf16sg.hip, tune 35) against the tiled kernel with the trainer's split-K rule (tune 36) on the
1x1 layers of the trainable trunk and their data gradients at 2 (and 1, 4) images: kernel time from a rocprofv3 trace (sg_trace.py lines the
launches up with the manifest) and bit equality against the UNSPLIT tiled launch."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from articulation3d_amd import ops

# name, H, W (input), Cin, Cout, stride, residual, gate
SH = [("res3 conv1 512->128", 60, 80, 512, 128, 1, False, False), ("res3 conv3 128->512 +res", 60, 80, 128, 512, 1, True, False),
      ("res3 dgrad conv3 512->128 gate", 60, 80, 512, 128, 1, False, True), ("res3 dgrad conv1 128->512 res+gate", 60, 80, 128, 512, 1, True, True),
      ("res4.0 conv1 512->256 s2", 60, 80, 512, 256, 2, False, False), ("res4 conv1 1024->256", 30, 40, 1024, 256, 1, False, False),
      ("res4 conv3 256->1024 +res", 30, 40, 256, 1024, 1, True, False), ("res4 dgrad conv3 1024->256 gate", 30, 40, 1024, 256, 1, False, True),
      ("res4 dgrad conv1 256->1024 res+gate", 30, 40, 256, 1024, 1, True, True), ("res4.0 shortcut 512->1024 s2", 60, 80, 512, 1024, 2, False, False),
      ("res5.0 conv1 1024->512 s2", 30, 40, 1024, 512, 2, False, False), ("res5 conv1 2048->512", 15, 20, 2048, 512, 1, False, False),
      ("res5 conv3 512->2048 +res", 15, 20, 512, 2048, 1, True, False), ("res5 dgrad conv3 2048->512 gate", 15, 20, 2048, 512, 1, False, True),
      ("res5 dgrad conv1 512->2048 res+gate", 15, 20, 512, 2048, 1, True, True), ("res5.0 shortcut 1024->2048 s2", 30, 40, 1024, 2048, 2, False, False),
      ("lateral5 2048->256 f32out", 15, 20, 2048, 256, 1, False, False), ("lateral4 1024->256", 30, 40, 1024, 256, 1, False, False),
      ("lateral3 512->256", 60, 80, 512, 256, 1, False, False)]
torch.manual_seed(0)
batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [2]
MANIFEST = []
bf = torch.bfloat16
for B in batches:
    for name, H, W, Cin, Cout, stride, res, gate in SH:
        x = torch.relu(torch.randn(B, H, W, Cin, device="cuda")).to(bf)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        r = torch.randn(B, Ho, Wo, Cout, device="cuda").to(bf) if res else None
        g = torch.randn(B, Ho, Wo, Cout, device="cuda").to(bf) if gate else None
        p = ops.pack_conv(torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5, torch.randn(Cout) * 0.1, None, stride, 0, ops.ACT_NONE if gate else ops.ACT_RELU)
        p.w_b16 = p.w.to(bf)
        od = None if "f32out" in name else bf
        call = lambda tune: ops.conv2d(x, p, res=r, gate=g, precision=1, out_dtype=od or torch.float32, tune=tune)
        ops.BF16_SPLITK_AUTO = False
        ref = call(36); vref = ops.last_conv_variant(); MANIFEST.append(("ref", 1))
        ops.BF16_SPLITK_AUTO = True
        out = []
        for tune in (36, 35, 0):
            y = call(tune); v = ops.last_conv_variant()
            eq = bool(torch.equal(y, ref))
            for _ in range(55): call(tune)
            nlaunch = 2 if " sk" in v else 1
            MANIFEST.append((f"B={B} {name} | tune {tune} {v}", 56 * nlaunch))
            out.append(f"{tune} {v:26s} {'eq' if eq else 'diff(split order)' if ' sk' in v else 'DIFF'}")
        M = B * Ho * Wo
        print(f"B={B} {name:38s} waves={((M + 31) // 32) * (Cout // 32):6d} ref {vref:20s} " + " | ".join(out), flush=True)
torch.cuda.synchronize()
json.dump(MANIFEST, open(os.environ.get("SG_MANIFEST", "/tmp/sg_manifest.json"), "w"))
