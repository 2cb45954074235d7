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


def _merge_expected(g, g0, ng, K):
    """Expected merge_topk output from the group buffers (CPU): kept entries by (score desc, pos asc)."""
    items = []
    for gl in range(ng):
        n = int(g["n"][g0 + gl])
        keep = g["keep"][g0 + gl, :n].bool()
        for r in keep.nonzero().squeeze(1).tolist():
            items.append((-float(g["scores"][g0 + gl, r]), int(g["pos"][g0 + gl, r]), gl, r))
    items.sort(key=lambda t: (t[0], t[1]))
    items = items[:K]
    boxes = torch.stack([g["boxes"][g0 + gl, r] for _, _, gl, r in items]) if items else torch.zeros(0, 4)
    return boxes, [it[2] for it in items]


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
    sizes = [(480, 640)] * nframes
    ok = True
    fr = torch.from_numpy(frames).to(dev)

    # ---- S1 continuous: backbone / RPN head / depth, end to end vs the CPU oracle
    x, _ = O.preprocess(O.frames_to_chw(frames), ocfg)
    ofeats = O.backbone(x, P)
    x4 = ops.preprocess_u8hwc(fr, model.pixel_mean, model.pixel_std)
    ok &= check("preprocess", x4[..., :3].permute(0, 3, 1, 2), x, 1e-7)
    feats = model.backbone.forward_nhwc(x4)
    names = ("p2", "p3", "p4", "p5", "p6")
    for k in names:
        ok &= check(f"feat {k}", feats[k].permute(0, 3, 1, 2), ofeats[k], 2e-4)
    rpn = model.proposal_generator
    heads = rpn.rpn_head.forward_nhwc([feats[f] for f in rpn.in_features])
    ol, od = O.rpn_head(ofeats, P)
    for l in range(5):
        ok &= check(f"rpn logits L{l}", heads[l][..., :3].reshape(nframes, -1), ol[l], 2e-4)
        ok &= check(f"rpn deltas L{l}", heads[l][..., 3:15].reshape(nframes, -1, 4), od[l], 2e-4)
    depth = model.depth_head.forward_nhwc(feats)
    ok &= check("depth e2e", depth, O.depth_head(ofeats, P), 5e-4)
    gfeats = {k: feats[k].permute(0, 3, 1, 2).contiguous().cpu() for k in names}  # GPU features as oracle input
    ok &= check("depth | same feats", depth, O.depth_head(gfeats, P), 2e-5)

    # ---- S2 discrete: proposal selection on IDENTICAL head outputs
    gl_ = [h[..., :3].reshape(nframes, -1).cpu() for h in heads]
    gd_ = [h[..., 3:15].reshape(nframes, -1, 4).cpu() for h in heads]
    feat_hw = [tuple(feats[k].shape[1:3]) for k in names]
    oprops, ogroups = O.rpn_select(gl_, gd_, feat_hw, sizes, ocfg, return_groups=True)
    pb, pl, plv, ppos, pc, g = rpn.forward_batched(feats, (480, 640), heads=heads, return_groups=True)
    g = {k: v.cpu() for k, v in g.items()}
    for b in range(nframes):
        for l in range(5):
            gi = b * 5 + l
            og = ogroups[b][l]
            k = len(og["scores"])
            same = int(g["n"][gi]) == k and torch.equal(g["scores"][gi, :k], og["scores"])
            eb = (g["boxes"][gi, :k] - og["boxes"]).abs().max().item()
            sv = torch.equal(g["valid"][gi, :k].bool(), og["valid"])
            v = g["valid"][gi, :k].bool()
            okeep = torch.zeros(k, dtype=torch.bool)
            okeep[v] = O.nms_sorted(g["boxes"][gi, :k][v], torch.zeros(int(v.sum()), dtype=torch.int64), ocfg.rpn_nms_thresh)
            sk = torch.equal(g["keep"][gi, :k].bool(), okeep)
            good = same and eb < 2e-3 and sv and sk
            print(f"{'OK ' if good else 'BAD'} rpn group b{b} L{l}: k={k} topk_scores_bitexact={same} box_abs_err={eb:.2e} valid_eq={sv} keep_bitexact={sk} kept={int(okeep.sum())}", flush=True)
            ok &= good
        eb_, _ = _merge_expected(g, b * 5, 5, 1000)
        n = int(pc[b])
        good = n == len(eb_) and torch.equal(pb[b, :n].cpu(), eb_)
        ob = oprops[b][0]
        loose = (n == len(ob)) and (pb[b, :n].cpu() - ob).abs().max().item()
        print(f"{'OK ' if good else 'BAD'} rpn merge b{b}: n={n} exact_vs_expected={good}; vs full CPU selection: n_oracle={len(ob)} max_box_diff={loose}", flush=True)
        ok &= good

    # ---- S3 box stage on IDENTICAL features + proposals
    rh = model.roi_heads
    lv = [feats[f] for f in rh.box_in_features]
    pooled = rh.box_pooler.forward_batched(lv, pb, pc)
    props_cpu = [pb[b, : int(pc[b])].cpu() for b in range(nframes)]
    opooled = O.roi_pool_fpn(gfeats, props_cpu, *ocfg.box_pool)
    R = pb.shape[1]
    gp = torch.cat([pooled[b * R: b * R + int(pc[b])] for b in range(nframes)]).permute(0, 3, 1, 2)
    ok &= check("box ROIAlign 7x7 | same feats", gp, opooled, 1e-5)
    fc = rh.box_head(pooled)
    pred = rh.box_predictor(fc)
    ofc = O.box_head(opooled, P)
    ocls, odl = O.box_predictor(ofc, P)
    gpred = torch.cat([pred[b * R: b * R + int(pc[b])] for b in range(nframes)])
    ok &= check("box cls logits", gpred[:, :3], ocls, 1e-4)
    ok &= check("box deltas", gpred[:, 3:11], odl, 1e-4)
    db, dsc, dcl, dpos, dcnt, g2 = rh.box_predictor.inference_batched(pred, pb, pc, (480, 640), return_groups=True)
    g2 = {k: v.cpu() for k, v in g2.items()}
    for b in range(nframes):
        n = int(pc[b])
        pr = pred[b * R: b * R + n].cpu()
        dec = O.apply_deltas(pr[:, 3:11], props_cpu[b], ocfg.box_weights, ocfg.scale_clamp)
        probs = F.softmax(pr[:, :3], dim=-1)
        ob_, os_, oc_, _rows = O.fast_rcnn_inference_single(dec, probs, (480, 640), ocfg)
        eb_, ecat = _merge_expected(g2, b * 2, 2, 100)
        nd = int(dcnt[b])
        exact = nd == len(eb_) and torch.equal(db[b, :nd].cpu(), eb_) and dcl[b, :nd].cpu().tolist() == ecat
        same_n = nd == len(ob_)
        ebx = (db[b, :nd].cpu() - ob_).abs().max().item() if same_n and nd else 0.0
        esc = (dsc[b, :nd].cpu() - os_).abs().max().item() if same_n and nd else 0.0
        ecl = (dcl[b, :nd].cpu().long() != oc_).sum().item() if same_n and nd else 0
        good = exact and same_n and ebx < 2e-3 and esc < 1e-6 and ecl == 0
        print(f"{'OK ' if good else 'BAD'} box det b{b}: D={nd} (oracle {len(ob_)}) merge_exact={exact} box_err={ebx:.2e} score_err={esc:.2e} class_mismatch={ecl}", flush=True)
        ok &= good
        for c in range(2):
            gi = b * 2 + c
            k = int(g2["n"][gi])
            v = g2["valid"][gi, :k].bool()
            okeep = torch.zeros(k, dtype=torch.bool)
            okeep[v] = O.nms_sorted(g2["boxes"][gi, :k][v], torch.zeros(int(v.sum()), dtype=torch.int64), ocfg.nms_thresh)
            sk = torch.equal(g2["keep"][gi, :k].bool(), okeep)
            print(f"{'OK ' if sk else 'BAD'} box nms b{b} c{c}: n={k} keep_bitexact={sk} kept={int(okeep.sum())}", flush=True)
            ok &= sk

    # ---- S4 per-ROI heads on IDENTICAL features + detection boxes
    from articulation3d_amd.modeling.roi_heads.roi_heads import BatchedDetections
    det = BatchedDetections(db, dsc, dcl, dcnt, (480, 640))
    det = rh.given_boxes_batched(feats, det)
    dets_cpu = [db[b, : int(dcnt[b])].cpu() for b in range(nframes)]
    if det.total:
        om = O.mask_head(O.roi_pool_fpn(gfeats, dets_cpu, *ocfg.mask_pool), P)
        ok &= check("mask probs", det.mask_prob[:, None], om, 1e-4)
        opl = O.plane_head(O.roi_pool_fpn(gfeats, dets_cpu, *ocfg.plane_pool), P)
        ok &= check("pred_plane", det.pred_plane, opl, 1e-4)
        ora, ota = O.axis_head(O.roi_pool_fpn(gfeats, dets_cpu, *ocfg.axis_pool), P)
        ok &= check("pred_rot_axis", det.pred_rot_axis, ora, 1e-4)
        ok &= check("pred_tran_axis", det.pred_tran_axis, ota, 1e-4)
    # ---- S5 post-process, paste, plane offset on IDENTICAL head outputs
    out = model._post_batched(det, depth, (480, 640), True, None)
    rays = O.k_inv_dot_xy1()
    for b in range(nframes):
        nd = int(dcnt[b])
        if nd == 0:
            print(f"OK  post b{b}: no detections")
            continue
        r0 = int(det.row_offset[b])
        d = dict(pred_boxes=db[b, :nd].cpu(), scores=dsc[b, :nd].cpu(), pred_classes=dcl[b, :nd].cpu().long(),
                 pred_masks=det.mask_prob[r0: r0 + nd, None].cpu(), pred_plane=det.pred_plane[r0: r0 + nd].cpu(), image_size=(480, 640))
        o = O.detector_postprocess(d, 480, 640, ocfg)
        keep = out.keep[b, :nd].bool().cpu()
        idx = keep.nonzero().squeeze(1)
        same = int(keep.sum()) == len(o["scores"])
        mm = (out.masks[b, idx].cpu().bool() != o["pred_masks"]).sum().item() if same else -1
        opo = O.override_depth(depth[b].cpu(), o["pred_masks"], o["pred_plane"], rays)
        epo = ((out.planes[b, idx].cpu() - opo).abs().max() / (opo.abs().max() + 1e-12)).item() if same else -1
        rc = int(out.rec_count[b])
        good = same and mm == 0 and epo < 1e-4 and rc == len(idx)
        print(f"{'OK ' if good else 'BAD'} post b{b}: kept={int(keep.sum())} (oracle {len(o['scores'])}) mask_px_mismatch={mm} of {len(idx) * 480 * 640} plane_offset_rel_err={epo:.2e} rec_count={rc}", flush=True)
        ok &= good
    # ---- whole path timing
    torch.cuda.synchronize()
    t0 = time.time()
    model.inference_batched(fr)
    torch.cuda.synchronize()
    print(f"hip inference_batched B={nframes}: {1e3 * (time.time() - t0):.1f} ms", flush=True)
    return ok


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), flush=True)
    ok = stage_conv()
    if "--conv-only" not in sys.argv:
        ok &= stage_model(2, 0.0)
        ok &= stage_model(2, 0.7)
    print("ALL OK" if ok else "SOME BAD")
    sys.exit(0 if ok else 1)
