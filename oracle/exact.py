"""Float64 yardstick for the end-to-end comparison (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Two fp32 evaluations of the same graph -- the reference's CPU operators and the HIP kernels -- differ by rounding, and the
PlaneRCNN graph amplifies that rounding on its way to the per-ROI outputs (50 backbone layers, then 6-layer heads ending in
an L2 normalisation of a 2- or 3-vector; on random-init weights the end-to-end deviation between two fp32 paths is 1e-4 ..
4e-3 of the output scale, measured on the MI355X: DESIGN.md section 4).  Which of the two is "right" can only be judged
against the EXACT evaluation, so this module evaluates the graph in float64 with the discrete choices of the fp32 oracle
run imposed on it (teacher forcing: same anchors kept as proposals, same (proposal, class) pairs kept as detections), which
makes every continuous output comparable rank for rank:

    err_cpu = | oracle fp32  - float64 |      (the reference path's own rounding error)
    err_hip = | HIP path     - float64 |

The end-to-end test asserts err_hip <= K * err_cpu per output (tests/test_gpu_e2e.py).

Same functions as planercnn_oracle.py (they are dtype-generic; ROIAlign has a double twin in a3d_oracle.c); follows
pkg/modeling/meta_arch/planercnn.py:148-219, pkg/modeling/roi_heads/roi_heads.py:167-273, pkg/utils/arti_vis.py:90-149.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from . import planercnn_oracle as O


def to_double(P: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}


def _pool(feats, box_lists, res, ratio, aligned):
    """ROIPooler (A.7) in float64."""
    names = ("p2", "p3", "p4", "p5")
    rois = torch.cat([torch.cat((torch.full((len(b), 1), float(i), dtype=torch.float64), b.double()), 1) for i, b in enumerate(box_lists)], 0)
    C = feats["p2"].shape[1]
    out = torch.zeros(rois.shape[0], C, res, res, dtype=torch.float64)
    if rois.shape[0] == 0:
        return out
    lv = O.assign_levels(rois[:, 1:])
    for li, name in enumerate(names):
        sel = (lv == li).nonzero().squeeze(1)
        if len(sel):
            out[sel] = O.roi_align(feats[name], rois[sel], res, 1.0 / O.FPN_STRIDES[name], ratio, aligned)
    return out


def _paste(masks, boxes, img_h, img_w, threshold):
    """mask_ops.py:41-60,128-129 in float64."""
    D = masks.shape[0]
    if D == 0:
        return torch.zeros(0, img_h, img_w, dtype=torch.bool)
    x0, y0, x1, y1 = torch.split(boxes, 1, dim=1)
    img_y = torch.arange(0, img_h, dtype=torch.float64) + 0.5
    img_x = torch.arange(0, img_w, dtype=torch.float64) + 0.5
    img_y = (img_y - y0) / (y1 - y0) * 2 - 1
    img_x = (img_x - x0) / (x1 - x0) * 2 - 1
    grid = torch.stack([img_x[:, None, :].expand(D, img_h, img_w), img_y[:, :, None].expand(D, img_h, img_w)], dim=3)
    return F.grid_sample(masks[:, None], grid, align_corners=False)[:, 0] >= threshold


def _override_depth(depth, masks, planes):
    """arti_vis.py:90-99,125-149 in float64 (rays: closed form of K^-1 [x, y, 1], arti_vis.py:101-123)."""
    h, w = depth.shape
    f = 571.623718
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float64), torch.arange(w, dtype=torch.float64), indexing="ij")
    rays = torch.stack([(xs - 319.5) / f, (ys - 239.5) / f, torch.ones_like(xs)], 0)
    xyz = rays * depth
    pl = planes.clone()
    pl[:, [1, 2]] = pl[:, [2, 1]]
    pl[:, 1] = -pl[:, 1]
    out = []
    for m, p in zip(masks, pl):
        if m.sum() == 0:
            out.append(p)
            continue
        n = p / p.norm().clamp(min=1e-8)
        out.append(n * (n @ xyz[:, m]).mean())
    if not out:
        return planes
    o = torch.stack(out)
    o[:, [1, 2]] = o[:, [2, 1]]
    o[:, 2] = -o[:, 2]
    return o


@torch.no_grad()
def detect_exact(images_chw: List[torch.Tensor], P64, cfg: O.OracleCfg, outs32: List[Dict], aux32: Dict) -> List[Dict]:
    """Float64 evaluation of the frames with the discrete selections of the fp32 run (`outs32, aux32 =
    planercnn_oracle.detect(..., return_aux=True)`) imposed.  Returns dicts with the keys of `detect`, rank-aligned with
    `outs32`."""
    names = ("p2", "p3", "p4", "p5", "p6")
    mean = torch.tensor(cfg.pixel_mean, dtype=torch.float64).view(-1, 1, 1)
    std = torch.tensor(cfg.pixel_std, dtype=torch.float64).view(-1, 1, 1)
    x = torch.stack([(im.double() - mean) / std for im in images_chw])
    H, W = x.shape[-2:]
    assert H % 32 == 0 and W % 32 == 0
    feats = O.backbone(x, P64)
    logits, deltas = O.rpn_head(feats, P64)
    props = []
    for n, (lv, ai) in enumerate(aux32["proposal_sources"]):
        b = torch.zeros(len(lv), 4, dtype=torch.float64)
        for li, name in enumerate(names):
            m = lv == li
            if m.any():
                Hf, Wf = feats[name].shape[-2:]
                anc = O.grid_anchors(Hf, Wf, O.FPN_STRIDES[name], cfg.anchor_sizes[li], cfg.anchor_ratios).double()
                b[m] = O.apply_deltas(deltas[li][n][ai[m]], anc[ai[m]], cfg.rpn_weights, cfg.scale_clamp)
        props.append(O.clip_boxes(b, H, W))
    pooled = _pool(feats, props, *cfg.box_pool)
    cls, dlt = O.box_predictor(O.box_head(pooled, P64), P64)
    probs = F.softmax(cls, dim=-1)
    nper = [len(p) for p in props]
    depth = O.depth_head(feats, P64)
    res = []
    boxes_all = []
    for n, (o32, pr, dl, pb) in enumerate(zip(outs32, probs.split(nper), dlt.split(nper), props)):
        rows, cl = o32["prop_rows"], o32["pred_classes"].long()
        dec = O.apply_deltas(dl[rows], pb[rows], cfg.box_weights, cfg.scale_clamp).view(len(rows), -1, 4)
        bx = O.clip_boxes(dec[torch.arange(len(rows)), cl], H, W)
        boxes_all.append(bx)
        res.append(dict(pred_boxes=bx, scores=pr[rows, cl], pred_classes=cl, depth=depth[n]))
    nd = [len(b) for b in boxes_all]
    m = O.mask_head(_pool(feats, boxes_all, *cfg.mask_pool), P64)
    pl, plr = O.plane_head(_pool(feats, boxes_all, *cfg.plane_pool), P64, return_raw=True)
    ra, ta, rar, tar = O.axis_head(_pool(feats, boxes_all, *cfg.axis_pool), P64, return_raw=True)
    for r, mm, pp, a, t, pr, ar, tr in zip(res, m.split(nd), pl.split(nd), ra.split(nd), ta.split(nd), plr.split(nd), rar.split(nd), tar.split(nd)):
        r["pred_plane"], r["pred_rot_axis"], r["pred_tran_axis"] = pp, a, t
        r["raw_plane"], r["raw_rot"], r["raw_tran"] = pr, ar, tr
        r["pred_masks"] = _paste(mm[:, 0], r["pred_boxes"], H, W, cfg.mask_threshold)
        r["plane_offset"] = _override_depth(r["depth"], r["pred_masks"], pp)
    return res
