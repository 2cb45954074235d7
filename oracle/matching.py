"""Checker for the north star's acceptance criterion, "matched detections": frames in -> detections out, HIP path vs
`planercnn_oracle.detect`, compared detection by detection; plus the margin of every discrete decision of the oracle's
run to its threshold / tie (SURVEY.md section 7 "hard parts", section 8d "matched detections").

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py, never by the product.

What is compared follows the reference's per-frame result (pkg/modeling/meta_arch/planercnn.py:125-184 -> `Instances`
with pred_boxes / scores / pred_classes / pred_masks / pred_plane / pred_rot_axis / pred_tran_axis + depth, then
pkg/utils/arti_vis.py:54-149 -> plane normal * offset).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from . import planercnn_oracle as O

# tolerances of the acceptance criterion (BASELINE.json north_star: "bit-exact box/class indices and NMS keep masks, plane
# normals and axis parameters within 1e-4 relative"); boxes pass through device expf vs libm expf (1 ulp of a <=640 px
# coordinate is 6e-5 px; the decode multiplies it by the box size), hence a pixel tolerance instead of bit equality.
# End to end (each path on its own upstream tensors) two fp32 evaluations of this graph differ by 3e-5 .. 9e-5 of the feature
# maximum after the backbone, and the softmax scores by up to 6e-5 (MI355X vs CPU, and CPU evaluation orders among
# themselves: oracle/seed_search.py); hence score 1e-4, and ranks whose scores are closer than 2*score count as tied.
TOL = dict(box_px=5e-3, score=1e-4, plane=1e-4, axis=1e-4, offset=1e-4, depth=1e-3)


def gpu_frame_results(out) -> List[Dict[str, torch.Tensor]]:
    """BatchedOutput of `PlaneRCNN.inference_batched(frames, want_masks=True)` -> per-frame dicts of CPU tensors with the
    keys of `planercnn_oracle.detect` (kept detections only, in the path's own order)."""
    cnt = out.det.count.tolist()
    keep = out.keep.bool().cpu()
    ro = out.det.row_offset.tolist() if out.det.total else [0] * (len(cnt) + 1)
    res = []
    for b, n in enumerate(cnt):
        idx = keep[b, :n].nonzero().squeeze(1)
        r = dict(pred_boxes=out.boxes[b].cpu()[idx], scores=out.det.scores[b].cpu()[idx],
                 pred_classes=out.det.classes[b].cpu()[idx].long(), depth=out.depth[b].cpu())
        if out.det.total:
            rows = ro[b] + idx
            r.update(pred_plane=out.det.pred_plane.cpu()[rows], pred_rot_axis=out.det.pred_rot_axis.cpu()[rows],
                     pred_tran_axis=out.det.pred_tran_axis.cpu()[rows], plane_offset=out.planes[b].cpu()[idx])
            if out.masks is not None:
                r["pred_masks"] = out.masks[b].cpu()[idx].bool()
            for k in ("raw_plane", "raw_rot", "raw_tran"):  # present when the heads ran with keep_raw (checker hook)
                if getattr(out.det, k, None) is not None:
                    r[k] = getattr(out.det, k).cpu()[rows]
        else:
            r.update(pred_plane=torch.zeros(0, 3), pred_rot_axis=torch.zeros(0, 3), pred_tran_axis=torch.zeros(0, 2),
                     plane_offset=torch.zeros(0, 3), pred_masks=torch.zeros(0, *out.image_size, dtype=torch.bool))
        res.append(r)
    return res


def _rel(a, b):
    if a.numel() == 0:
        return 0.0
    return float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12))


def _align(g: Dict, o: Dict, tie: float):
    """Rank order is defined by the score, and the two paths' scores differ by fp32 rounding: detections whose ORACLE
    scores are within `tie` of each other are tied, and the paths may list them in either order.  Returns the permutation
    perm (oracle rank r <-> HIP rank perm[r]) that pairs every oracle detection with the nearest same-class HIP box, or
    None when that is not a bijection or when it moves a detection across a score gap larger than `tie`."""
    n = len(o["scores"])
    if n == 0:
        return torch.zeros(0, dtype=torch.int64)
    d = (o["pred_boxes"][:, None, :] - g["pred_boxes"][None, :, :]).abs().amax(-1)
    d = d + 1e6 * (o["pred_classes"].long()[:, None] != g["pred_classes"].long()[None, :])
    perm = d.argmin(1)
    if len(set(perm.tolist())) != n:
        return None
    so = o["scores"]
    moved = (perm != torch.arange(n)).nonzero().squeeze(1)
    for r in moved.tolist():
        if abs(float(so[r] - so[int(perm[r])])) > tie:
            return None
    return perm


def compare_frame(g: Dict, o: Dict) -> Dict:
    """One frame: HIP result `g` vs oracle result `o` (dicts as returned by `detect`).  `matched` = same number of
    detections, the same detections rank for rank (class index and box; ranks may be exchanged only inside a group of
    scores tied to within 2*TOL['score'], see _align), boxes and scores within TOL; the deviations of the continuous
    per-ROI outputs are reported next to it."""
    m = dict(n_gpu=len(g["scores"]), n_cpu=len(o["scores"]))
    m["same_count"] = m["n_gpu"] == m["n_cpu"]
    if not m["same_count"]:
        m["matched"] = False
        return m
    n = m["n_cpu"]
    perm = _align(g, o, 2 * TOL["score"])
    m["classes_equal"] = perm is not None
    if perm is None:  # not the same detections: report the rank-for-rank deviation of the raw lists
        m["box_err_px"] = float((g["pred_boxes"] - o["pred_boxes"]).abs().max())
        m["ranks_differing"] = int(((g["pred_boxes"] - o["pred_boxes"]).abs().amax(1) > TOL["box_px"]).sum())
        m["matched"] = False
        return m
    m["tied_rank_swaps"] = int((perm != torch.arange(n)).sum())
    g = {k: (v[perm] if torch.is_tensor(v) and k != "depth" and v.shape[:1] == (n,) else v) for k, v in g.items()}
    m["box_err_px"] = float((g["pred_boxes"] - o["pred_boxes"]).abs().max()) if n else 0.0
    m["score_err"] = float((g["scores"] - o["scores"]).abs().max()) if n else 0.0
    m["plane_rel"] = _rel(g["pred_plane"], o["pred_plane"])
    m["rot_axis_rel"] = _rel(g["pred_rot_axis"], o["pred_rot_axis"])
    m["tran_axis_rel"] = _rel(g["pred_tran_axis"], o["pred_tran_axis"])
    # the same three, detection by detection (max |difference| of a detection / the frame's scale): the per-ROI outputs are
    # normalised vectors, so a detection whose raw vector is short amplifies rounding -- their maximum over hundreds of
    # detections is heavy-tailed, and tests compare quantiles of these lists rather than single maxima
    for key, name in (("pred_plane", "plane"), ("pred_rot_axis", "rot_axis"), ("pred_tran_axis", "tran_axis")):
        d = (g[key].float() - o[key].float()).abs().flatten(1).amax(1) / (o[key].float().abs().max() + 1e-12) if n else torch.zeros(0)
        m[name + "_err_all"] = [float(v) for v in d]
    # RAW head vectors (before F.normalize) and the normalised outputs weighted by their conditioning.  n = r / |r| has
    # dn ~ dr / |r|: the whole heavy tail of the normalised errors is the factor 1 / |r| of short raw vectors, so
    #   raw_*_err_all   = |r_g - r_o|_inf per detection (no amplification: comparable at its MAXIMUM), and
    #   *_cond_all      = |n_g - n_o|_inf * |r_o|       (the normalised error with the amplification divided out)
    # are held to the float64 yardstick at their maxima (tests/test_gpu_e2e.py); both in units of the frame's largest |r_o|.
    for raw, key, name, nn_ in (("raw_plane", "pred_plane", "plane", 3), ("raw_rot", "pred_rot_axis", "rot_axis", 2), ("raw_tran", "pred_tran_axis", "tran_axis", 2)):
        if raw in g and raw in o and n:
            ro, rg = o[raw].double(), g[raw].double()
            scale = float(ro.abs().max()) + 1e-300
            m[raw + "_rel"] = float((rg - ro).abs().max() / scale)
            m[raw + "_err_all"] = [float(v) for v in (rg - ro).abs().amax(1) / scale]
            nrm = ro[:, :nn_].norm(dim=1)
            dn = (g[key].double()[:, :nn_] - o[key].double()[:, :nn_]).abs().amax(1)
            m[name + "_cond_all"] = [float(v) for v in dn * nrm / scale]
            m[name + "_rawnorm_all"] = [float(v) for v in nrm / scale]
    same_mask = torch.ones(n, dtype=torch.bool)
    if "pred_masks" in g and "pred_masks" in o and n:
        diff = (g["pred_masks"] != o["pred_masks"]).flatten(1).sum(1)
        area = o["pred_masks"].flatten(1).sum(1).clamp(min=1)
        m["mask_hamming_px"] = int(diff.sum())
        m["mask_hamming_max_frac"] = float((diff.float() / area.float()).max())
        m["masks_differing"] = int((diff > 0).sum())
        same_mask = diff == 0
    else:
        m["mask_hamming_px"], m["mask_hamming_max_frac"], m["masks_differing"] = 0, 0.0, 0
    if "plane_offset" in o and "plane_offset" in g:
        # the offset is a mean over the pasted mask (arti_vis.py:139): a mask pixel that sits within rounding of the 0.5
        # threshold flips it discontinuously, so the continuous comparison runs over the detections whose masks agree
        # (the others are counted in masks_differing)
        m["plane_offset_rel"] = _rel(g["plane_offset"][same_mask], o["plane_offset"][same_mask]) if bool(same_mask.any()) else 0.0
    if o.get("depth") is not None and g.get("depth") is not None:
        m["depth_rel"] = _rel(g["depth"], o["depth"])
    # "matched detections" (SURVEY.md 8d): identical D, the same detections rank for rank, boxes (and scores) within TOL
    m["matched"] = bool(m["box_err_px"] <= TOL["box_px"] and m["score_err"] <= TOL["score"])
    # the per-ROI head outputs at the flat 1e-4 of the north star (holds on identical inputs; end to end see oracle/exact.py)
    m["heads_within_1e-4"] = bool(m["plane_rel"] <= TOL["plane"] and m["rot_axis_rel"] <= TOL["axis"] and m["tran_axis_rel"] <= TOL["axis"]
                                  and m.get("plane_offset_rel", 0.0) <= TOL["offset"])
    return m


def summarize(ms: List[Dict]) -> Dict:
    """Aggregate of per-frame comparisons (what bench.py prints and the test reports)."""
    keys = ("box_err_px", "score_err", "plane_rel", "rot_axis_rel", "tran_axis_rel", "plane_offset_rel", "depth_rel", "mask_hamming_max_frac")
    s = dict(frames=len(ms), matched_frames=sum(1 for m in ms if m["matched"]), matched=all(m["matched"] for m in ms),
             frames_heads_within_1e4=sum(1 for m in ms if m.get("heads_within_1e-4", False)),
             detections=[m["n_cpu"] for m in ms], detections_gpu=[m["n_gpu"] for m in ms])
    for k in keys:
        v = [m[k] for m in ms if k in m]
        if v:
            s["max_" + k] = float(max(v))
    s["mask_hamming_px"] = int(sum(m.get("mask_hamming_px", 0) for m in ms))
    for k in ("raw_plane_err_all", "raw_rot_err_all", "raw_tran_err_all", "plane_cond_all", "rot_axis_cond_all", "tran_axis_cond_all"):
        v = [x for m in ms for x in m.get(k, [])]
        if v:
            s["max_" + k[:-4]] = float(max(v))
    return s


# ------------------------------------------------------------------------------------------------ margins
def _pair_iou(b):
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(b[:, None, :2], b[None, :, :2])
    rb = torch.min(b[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area[:, None] + area[None, :] - inter)


def nms_margin(boxes, cats, keep, thr) -> float:
    """boxes score-descending, `keep` the NMS keep mask on them.  For every candidate j the decision is
    max_{kept i<j, same category} IoU(i,j) > thr; the margin is the smallest |that maximum - thr| over j (candidates with
    no kept overlapping predecessor have margin thr).  A perturbation of the IoUs below this margin cannot change the keep
    mask."""
    n = len(boxes)
    if n < 2:
        return float(thr)
    iou = _pair_iou(boxes.double())
    same = cats[:, None] == cats[None, :]
    earlier_kept = torch.tril(torch.ones(n, n, dtype=torch.bool), -1) & keep[None, :]
    iou = torch.where(same & earlier_kept, iou, torch.zeros((), dtype=iou.dtype))
    mx = iou.max(1).values
    return float((mx - thr).abs().min())


def _gap_at(sorted_desc: torch.Tensor, k: int) -> float:
    """score gap between rank k-1 (last taken) and rank k (first dropped); inf if nothing is dropped."""
    if len(sorted_desc) <= k or k == 0:
        return float("inf")
    return float(sorted_desc[k - 1] - sorted_desc[k])


def _level_margin(boxes) -> float:
    """distance of 4 + log2(sqrt(area)/224 + 1e-8) to the nearest level boundary that matters (3, 4, 5: levels clamp to
    [2, 5]), A.7."""
    if len(boxes) == 0:
        return float("inf")
    area = ((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])).double()
    v = 4 + torch.log2(torch.sqrt(area) / 224 + 1e-8)
    return float(torch.stack([(v - t).abs() for t in (3.0, 4.0, 5.0)]).min())


@torch.no_grad()
def decision_margins(feats, P, cfg: O.OracleCfg, image_size=(480, 640)) -> Dict[str, float]:
    """Margins of the oracle's own discrete decisions on ONE frame (features [1,C,H,W] per level), in the unit of the
    quantity that is thresholded: logits for the RPN rank cuts, IoU for the two NMS stages, probability for the detection
    score threshold / rank-100 cut / order, levels (log2 units) for the FPN level assignment.  `min_effective` keeps only
    the decisions behind the final detections (box stage); the RPN-stage margins are reported separately because a flip
    there changes one proposal of 1000 and usually no detection."""
    names = ("p2", "p3", "p4", "p5", "p6")
    logits, deltas = O.rpn_head(feats, P)
    feat_hw = [tuple(feats[n].shape[-2:]) for n in names]
    props, groups = O.rpn_select(logits, deltas, feat_hw, [image_size], cfg, return_groups=True)
    g = groups[0]
    m = {}
    m["rpn_level_topk_gap"] = min(_gap_at(torch.sort(logits[l][0], descending=True).values, cfg.rpn_pre_topk) for l in range(5))
    b = torch.cat([x["boxes"][x["valid"]] for x in g])
    s = torch.cat([x["scores"][x["valid"]] for x in g])
    lv = torch.cat([torch.full((int(x["valid"].sum()),), i) for i, x in enumerate(g)])
    order = torch.sort(s, descending=True, stable=True)[1]
    keep = O.nms_sorted(b[order], lv[order], cfg.rpn_nms_thresh)
    m["rpn_nms_iou_margin"] = nms_margin(b[order], lv[order], keep, cfg.rpn_nms_thresh)
    kept_scores = s[order][keep]
    m["rpn_post_topk_gap"] = _gap_at(kept_scores, cfg.rpn_post_topk)
    pb = props[0][0]
    m["box_pooler_level_margin"] = _level_margin(pb)
    # box stage
    pooled = O.roi_pool_fpn(feats, [pb], *cfg.box_pool)
    cls, dlt = O.box_predictor(O.box_head(pooled, P), P)
    dec = O.apply_deltas(dlt, pb, cfg.box_weights, cfg.scale_clamp)
    probs = F.softmax(cls, dim=-1)[:, :-1]
    C = probs.shape[1]
    boxes = O.clip_boxes(dec.reshape(-1, 4), *image_size).view(-1, C, 4)
    m["score_thresh_margin"] = float((probs - cfg.score_thresh).abs().min())
    fmask = probs > cfg.score_thresh
    finds = fmask.nonzero()
    cb, cs, cc = boxes[fmask], probs[fmask], finds[:, 1]
    order = torch.sort(cs, descending=True, stable=True)[1]
    keep = O.nms_sorted(cb[order], cc[order], cfg.nms_thresh)
    m["det_nms_iou_margin"] = nms_margin(cb[order], cc[order], keep, cfg.nms_thresh) if len(cb) else float("inf")
    ks = cs[order][keep]
    m["det_topk_gap"] = _gap_at(ks, cfg.dets_per_image)
    top = ks[: cfg.dets_per_image]
    m["det_order_gap"] = float((top[:-1] - top[1:]).min()) if len(top) > 1 else float("inf")
    m["post_score_margin"] = float((top - cfg.post_score_thresh).abs().min()) if len(top) else float("inf")
    m["det_pooler_level_margin"] = _level_margin(cb[order][keep][: cfg.dets_per_image])
    m["detections"] = int(min(len(ks), cfg.dets_per_image))
    return m


def same_discrete_result(a: Dict, b: Dict, box_px=0.05) -> bool:
    """Two oracle results describe the same detections: count, and rank for rank (up to score ties, _align) the same
    class and the same box within `box_px`."""
    if len(a["scores"]) != len(b["scores"]):
        return False
    perm = _align(b, a, 2 * TOL["score"])
    if perm is None:
        return False
    return len(a["scores"]) == 0 or float((b["pred_boxes"][perm] - a["pred_boxes"]).abs().max()) <= box_px
