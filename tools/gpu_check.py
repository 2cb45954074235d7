"""Developer bring-up script (run on the GPU box through gpurun): staged HIP-vs-oracle comparisons with
printed error figures.  The formal versions of these checks live in tests/ (-m gpu)."""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from articulation3d_amd import ops  # noqa: E402

dev = "cuda"
torch.manual_seed(0)


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def check(name, got, ref, tol=2e-5):
    e = rel(got, ref)
    print(f"{'OK ' if e < tol else 'BAD'} {name}: rel_max_err={e:.3e} shape={tuple(ref.shape)}", flush=True)
    return e < tol


def stage_conv():
    ok = True
    cases = [
        ("1x1 64->256", dict(B=2, H=24, W=40, Cin=64, Cout=256, k=1, s=1, p=0)),
        ("1x1 256->64", dict(B=2, H=24, W=40, Cin=256, Cout=64, k=1, s=1, p=0)),
        ("3x3 64->64", dict(B=2, H=24, W=40, Cin=64, Cout=64, k=3, s=1, p=1)),
        ("3x3 256->256 odd", dict(B=3, H=15, W=20, Cin=256, Cout=256, k=3, s=1, p=1)),
        ("1x1 s2 256->512", dict(B=2, H=24, W=40, Cin=256, Cout=512, k=1, s=2, p=0)),
        ("1x1 256->15", dict(B=2, H=8, W=10, Cin=256, Cout=15, k=1, s=1, p=0)),
        ("3x3 128->128 big", dict(B=1, H=60, W=80, Cin=128, Cout=128, k=3, s=1, p=1)),
    ]
    for name, c in cases:
        x = torch.randn(c["B"], c["Cin"], c["H"], c["W"])
        w = torch.randn(c["Cout"], c["Cin"], c["k"], c["k"]) / (c["Cin"] * c["k"] ** 2) ** 0.5
        b = torch.randn(c["Cout"])
        ref = F.relu(F.conv2d(x, w, b, stride=c["s"], padding=c["p"]))
        p = ops.pack_conv(w, b, None, c["s"], c["p"], ops.ACT_RELU, device=dev)
        y = ops.conv2d(nhwc(x).to(dev), p)
        ok &= check("conv " + name, y[..., : c["Cout"]].permute(0, 3, 1, 2), ref)
    # BN fold + residual
    x = torch.randn(2, 64, 24, 40)
    w = torch.randn(256, 64, 1, 1) / 8
    bn = (torch.rand(256) + 0.5, torch.randn(256) * 0.1, torch.randn(256) * 0.1, torch.rand(256) + 0.5, 1e-5)
    r = torch.randn(2, 256, 24, 40)
    ref = F.relu(F.batch_norm(F.conv2d(x, w), bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5) + r)
    p = ops.pack_conv(w, None, bn, 1, 0, ops.ACT_RELU, device=dev)
    y = ops.conv2d(nhwc(x).to(dev), p, res=nhwc(r).to(dev))
    ok &= check("conv bn+res", y.permute(0, 3, 1, 2), ref)
    # res_ups (FPN lateral)
    x = torch.randn(2, 512, 30, 40)
    w = torch.randn(256, 512, 1, 1) / 22
    b = torch.randn(256)
    prev = torch.randn(2, 256, 15, 20)
    ref = F.conv2d(x, w, b) + F.interpolate(prev, scale_factor=2.0, mode="nearest")
    p = ops.pack_conv(w, b, None, 1, 0, device=dev)
    y = ops.conv2d(nhwc(x).to(dev), p, res=nhwc(prev).to(dev), res_ups=True)
    ok &= check("conv res_ups", y.permute(0, 3, 1, 2), ref)
    # upsample + concat (depth deconv)
    a = torch.randn(2, 128, 15, 20)
    c2 = torch.randn(2, 128, 15, 20)
    w = torch.randn(128, 256, 3, 3) / 48
    b = torch.randn(128)
    ref = F.relu(F.conv2d(F.interpolate(torch.cat([a, c2], 1), scale_factor=2, mode="nearest"), w, b, padding=1))
    p = ops.pack_conv(w, b, None, 1, 1, ops.ACT_RELU, device=dev)
    y = ops.conv2d(nhwc(a).to(dev), p, x2=nhwc(c2).to(dev), ups=True)
    ok &= check("conv ups+cat", y.permute(0, 3, 1, 2), ref)
    # stem
    x = torch.rand(2, 3, 96, 128) * 255 - 110
    w = torch.randn(64, 3, 7, 7) / 12
    bn = (torch.rand(64) + 0.5, torch.randn(64) * 0.1, torch.randn(64) * 0.1, torch.rand(64) + 0.5, 1e-5)
    ref = F.relu(F.batch_norm(F.conv2d(x, w, None, 2, 3), bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5))
    p = ops.pack_stem(w, bn, device=dev)
    x4 = ops.preprocess_f32chw(x.to(dev), (0, 0, 0), (1, 1, 1))
    y = ops.conv2d(x4, p)
    ok &= check("conv stem", y.permute(0, 3, 1, 2), ref)
    ok &= check("maxpool", ops.maxpool3x3s2(y).permute(0, 3, 1, 2), F.max_pool2d(ref, 3, 2, 1))
    # deconv 2x2
    x = torch.randn(5, 256, 14, 14)
    w = torch.randn(256, 256, 2, 2) / 16
    b = torch.randn(256)
    ref = F.relu(F.conv_transpose2d(x, w, b, stride=2))
    p = ops.pack_deconv2x2(w, b, device=dev)
    y = ops.conv2d(nhwc(x).to(dev), p)
    ok &= check("deconv2x2", y.permute(0, 3, 1, 2), ref)
    # linear chw + splitk
    x = torch.randn(7, 256, 14, 14)
    w = torch.randn(1024, 256 * 14 * 14) / 224
    b = torch.randn(1024)
    ref = F.relu(F.linear(x.flatten(1), w, b))
    p = ops.pack_linear(w, b, chw=(256, 14, 14), act=ops.ACT_RELU, device=dev)
    xr = nhwc(x).reshape(7, -1).to(dev)
    ok &= check("linear 50176 sk1", ops.linear(xr, p), ref)
    ok &= check("linear 50176 sk16", ops.linear(xr, p, splitk=16), ref)
    ok &= check("linear 50176 sk-auto", ops.linear(xr, p, splitk=ops.choose_splitk(7, 1024, 50176)), ref)
    # resize + conv3x3_to1
    x = torch.randn(2, 128, 16, 20)
    ok &= check("resize 16x20->15x20", ops.resize_bilinear(nhwc(x).to(dev), 15, 20).permute(0, 3, 1, 2),
                F.interpolate(x, size=(15, 20), mode="bilinear", align_corners=False))
    x = torch.randn(2, 1, 24, 32)
    ok &= check("resize x2 C1", ops.resize_bilinear(nhwc(x).to(dev), 48, 64).permute(0, 3, 1, 2),
                F.interpolate(x, size=(48, 64), mode="bilinear", align_corners=False))
    x = torch.randn(2, 64, 24, 32)
    w = torch.randn(1, 64, 3, 3) / 24
    ok &= check("conv3x3_to1", ops.conv3x3_to1(nhwc(x).to(dev), w[0].permute(1, 2, 0).contiguous().to(dev), 0.3)[:, None],
                F.conv2d(x, w, torch.tensor([0.3]), padding=1))
    return ok


def stage_model(nframes=2, thresh=0.0):
    from oracle import planercnn_oracle as O
    from articulation3d_amd.config import get_cfg, get_planercnn_cfg_defaults
    from articulation3d_amd.modeling import build_model

    t0 = time.time()
    P = O.init_params(2020)
    print(f"oracle params {time.time() - t0:.1f}s", flush=True)
    cfg = get_cfg()
    get_planercnn_cfg_defaults(cfg)
    cfg.merge_from_file(os.path.join(os.path.dirname(__file__), "..", "configs", "planercnn_inference.yaml"))
    cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = thresh
    model = build_model(cfg).eval()
    missing, unexpected = model.load_state_dict(P, strict=False)
    assert not unexpected and all("num_batches_tracked" in k for k in missing), (missing, unexpected)
    frames = O.synthetic_frames(nframes)
    ocfg = O.OracleCfg(score_thresh=thresh)
    t0 = time.time()
    outs, aux = O.detect(O.frames_to_chw(frames), P, ocfg, return_aux=True)
    print(f"oracle detect {time.time() - t0:.1f}s D={[len(o['scores']) for o in outs]}", flush=True)
    fr = torch.from_numpy(frames).to(dev)
    torch.cuda.synchronize()
    t0 = time.time()
    out = model.inference_batched(fr, want_masks=True)
    torch.cuda.synchronize()
    print(f"hip detect (cold) {time.time() - t0:.2f}s", flush=True)
    ok = True
    # features
    x4 = ops.preprocess_u8hwc(fr, model.pixel_mean, model.pixel_std)
    feats = model.backbone.forward_nhwc(x4)
    for k in ("p2", "p3", "p4", "p5", "p6"):
        ok &= check(f"feat {k}", feats[k].permute(0, 3, 1, 2), aux["features"][k], 1e-4)
    # proposals
    pb, pl, plv, ppos, pc = out.proposals
    for b in range(nframes):
        ob, osc = aux["proposals"][b]
        n = int(pc[b])
        same_n = n == len(ob)
        if same_n:
            e = (pb[b, :n].cpu() - ob).abs().max().item()
            es = (pl[b, :n].cpu() - osc).abs().max().item()
        else:
            e = es = float("nan")
        print(f"{'OK ' if same_n and e < 1e-2 else 'BAD'} proposals[{b}] n={n} vs {len(ob)} box_abs_err={e:.3e} logit_err={es:.3e}")
        ok &= same_n and e < 1e-2
    # depth
    ok &= check("depth", out.depth, torch.stack([o["depth"] for o in outs]), 1e-4)
    # detections
    det = out.det
    for b in range(nframes):
        o = outs[b]
        n = int(det.count[b])
        keep = out.keep[b, :n].bool().cpu()
        nk = int(keep.sum())
        good = nk == len(o["scores"])
        msg = f"det[{b}] raw={n} kept={nk} vs oracle {len(o['scores'])}"
        if good and nk:
            idx = keep.nonzero().squeeze(1)
            eb = (out.boxes[b, idx].cpu() - o["pred_boxes"]).abs().max().item()
            es = (det.scores[b, idx].cpu() - o["scores"]).abs().max().item()
            ec = (det.classes[b, idx].cpu().long() != o["pred_classes"]).sum().item()
            rows = (det.row_offset[b].item() + idx)
            epl = (det.pred_plane[rows].cpu() - o["pred_plane"]).abs().max().item()
            era = (det.pred_rot_axis[rows].cpu() - o["pred_rot_axis"]).abs().max().item()
            eta = (det.pred_tran_axis[rows].cpu() - o["pred_tran_axis"]).abs().max().item()
            mm = (out.masks[b, idx].cpu().bool() != o["pred_masks"]).sum().item()
            epo = ((out.planes[b, idx].cpu() - o["plane_offset"]).abs().max() / (o["plane_offset"].abs().max() + 1e-12)).item()
            msg += f" box={eb:.2e} score={es:.2e} cls_mismatch={ec} plane={epl:.2e} rot={era:.2e} tran={eta:.2e} mask_px_mismatch={mm} plane_off_rel={epo:.2e}"
            good = eb < 1e-2 and es < 1e-4 and ec == 0 and epl < 1e-4 and era < 1e-3 and eta < 1e-4 and epo < 1e-3
        print(("OK  " if good else "BAD ") + msg, flush=True)
        ok &= good
    return ok


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), flush=True)
    ok = stage_conv()
    if "--conv-only" not in sys.argv:
        ok &= stage_model(2, 0.0)
        ok &= stage_model(2, 0.7)
    print("ALL OK" if ok else "SOME BAD")
    sys.exit(0 if ok else 1)
