"""CPU oracle of the detector TRAINING step -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and tools/train_bench.py's cpu_baseline leg may import this module; the
product (articulation3d_amd/) never does.

What it restates: the branch `PlaneRCNN.forward` takes when `self.training` (reference
articulation3d/articulation3d/modeling/meta_arch/planercnn.py:83-123) under config/step1_bbox.yaml (MASK/DEPTH/PLANE/
AXIS off, so `detector_losses` = the box branch only: roi_heads.py:93-117 -> `_forward_box` :190-204) with the solver of
tools/train_net.py:84-117 (detectron2 DefaultTrainer -> SGD + WarmupMultiStepLR).  Everything numerical below that
boundary is detectron2 code that is NOT vendored under /root/reference (setup.py:10, unpinned): RPN anchor labelling
and losses, ROI sampling, FastRCNNOutputLayers.losses, Box2BoxTransform.get_deltas, Matcher, subsample_labels, SGD.
They are restated here from detectron2 v0.6's published algorithm; the reference holds no test or golden vector for
any of them -> PARITY UNPINNED for the training step; this file is the parity definition.  Gradients come from
torch.autograd over the forward oracle (oracle/planercnn_oracle.py), ROIAlign through the C restatement
(oracle/a3d_oracle.c: orc_roi_align_nchw / orc_roi_align_backward_nchw).

Random sampling (subsample_labels uses torch.randperm) is made reproducible by drawing from an explicit CPU
generator; parity tests feed the SAME sampled index sets to the HIP step.
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import planercnn_oracle as O


@dataclass
class TrainCfg:
    # d2 defaults selected by config/step1_bbox.yaml (its _BASE_ line is commented out -> detectron2 defaults)
    rpn_batch_per_image: int = 256  # MODEL.RPN.BATCH_SIZE_PER_IMAGE
    rpn_positive_fraction: float = 0.5  # MODEL.RPN.POSITIVE_FRACTION
    rpn_iou_thresholds: Tuple[float, float] = (0.3, 0.7)  # MODEL.RPN.IOU_THRESHOLDS, labels [0,-1,1]
    rpn_pre_topk_train: int = 2000  # step1_bbox.yaml:21
    rpn_post_topk_train: int = 1000  # step1_bbox.yaml:26
    roi_batch_per_image: int = 512  # MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE
    roi_positive_fraction: float = 0.25  # MODEL.ROI_HEADS.POSITIVE_FRACTION
    roi_iou_threshold: float = 0.5  # MODEL.ROI_HEADS.IOU_THRESHOLDS [0.5], labels [0,1]
    proposal_append_gt: bool = True  # MODEL.ROI_HEADS.PROPOSAL_APPEND_GT
    smooth_l1_beta: float = 0.0  # MODEL.RPN.SMOOTH_L1_BETA / ROI_BOX_HEAD.SMOOTH_L1_BETA (-> plain L1)
    base_lr: float = 0.001  # SOLVER.BASE_LR
    momentum: float = 0.9
    weight_decay: float = 1e-4  # weights and biases alike (WEIGHT_DECAY_BIAS = WEIGHT_DECAY; norms are frozen)
    warmup_iters: int = 1000
    warmup_factor: float = 0.001
    steps: Tuple[int, ...] = (210000, 250000)  # step1_bbox.yaml:37
    gamma: float = 0.1


FROZEN_PREFIXES = ("backbone.bottom_up.stem.", "backbone.bottom_up.res2.")  # MODEL.BACKBONE.FREEZE_AT 2


def trainable_names(P: Dict[str, torch.Tensor]) -> List[str]:
    """Parameters the step1 solver updates: all conv / linear weights and biases of res3-5, FPN, RPN head, box head and
    predictor.  FrozenBN statistics are buffers, stem + res2 are frozen, the other heads are switched off."""
    keep = []
    for k in P:
        if ".norm." in k or k.startswith(FROZEN_PREFIXES):
            continue
        if k.startswith(("backbone.", "proposal_generator.", "roi_heads.box_head.", "roi_heads.box_predictor.")):
            keep.append(k)
    return keep


# ----------------------------------------------------------------------------------------------
# detectron2 structures / matcher / sampling
# ----------------------------------------------------------------------------------------------
def pairwise_iou(b1: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """detectron2.structures.pairwise_iou: [M,4] x [N,4] -> [M,N]; zero where the intersection is empty."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    wh = torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])
    wh.clamp_(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (a1[:, None] + a2 - inter), torch.zeros(1, dtype=inter.dtype))


def matcher(mqm: torch.Tensor, thresholds, labels, allow_low_quality: bool):
    """detectron2.modeling.matcher.Matcher.__call__ on a [G, N] quality matrix -> (matched gt index [N], label int8 [N])."""
    N = mqm.shape[1]
    if mqm.numel() == 0:
        return torch.zeros(N, dtype=torch.int64), torch.full((N,), labels[0], dtype=torch.int8)
    vals, matches = mqm.max(dim=0)
    out = torch.full((N,), 1, dtype=torch.int8)
    th = [-float("inf")] + list(thresholds) + [float("inf")]
    for l, lo, hi in zip(labels, th[:-1], th[1:]):
        out[(vals >= lo) & (vals < hi)] = l
    if allow_low_quality:
        best, _ = mqm.max(dim=1)
        out[(mqm == best[:, None]).nonzero()[:, 1]] = 1
    return matches, out


def subsample_labels(labels: torch.Tensor, num: int, pos_frac: float, bg_label: int, gen: torch.Generator):
    """detectron2.modeling.sampling.subsample_labels with an explicit CPU generator."""
    positive = ((labels != -1) & (labels != bg_label)).nonzero().squeeze(1)
    negative = (labels == bg_label).nonzero().squeeze(1)
    num_pos = min(positive.numel(), int(num * pos_frac))
    num_neg = min(negative.numel(), num - num_pos)
    p1 = torch.randperm(positive.numel(), generator=gen)[:num_pos]
    p2 = torch.randperm(negative.numel(), generator=gen)[:num_neg]
    return positive[p1], negative[p2]


def get_deltas(src: torch.Tensor, tgt: torch.Tensor, weights) -> torch.Tensor:
    """Box2BoxTransform.get_deltas."""
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
    tx, ty = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
    wx, wy, ww, wh = weights
    return torch.stack((wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)), dim=1)


def all_anchors(feat_hw, cfg: O.OracleCfg) -> torch.Tensor:
    names = ("p2", "p3", "p4", "p5", "p6")
    return torch.cat([O.grid_anchors(h, w, O.FPN_STRIDES[n], cfg.anchor_sizes[i], cfg.anchor_ratios)
                      for i, (n, (h, w)) in enumerate(zip(names, feat_hw))], 0)


def match_anchors(anchors, gt_boxes: List[torch.Tensor], tc: TrainCfg):
    """First half of RPN.label_and_sample_anchors: per image (matched gt index [A], label int8 [A] in {-1,0,1})."""
    out = []
    for gb in gt_boxes:
        out.append(matcher(pairwise_iou(gb, anchors), tc.rpn_iou_thresholds, (0, -1, 1), True))
    return out


def sample_anchors(labels: torch.Tensor, tc: TrainCfg, gen) -> torch.Tensor:
    """RPN._subsample_labels: keep 256 anchors (<= half positive), the rest become -1."""
    pos, neg = subsample_labels(labels, tc.rpn_batch_per_image, tc.rpn_positive_fraction, 0, gen)
    out = torch.full_like(labels, -1)
    out[pos] = 1
    out[neg] = 0
    return out


def rpn_losses(logits: List[torch.Tensor], deltas: List[torch.Tensor], anchors, labels: torch.Tensor,
               matched_gt: torch.Tensor, cfg: O.OracleCfg, tc: TrainCfg):
    """RPN.losses.  logits[l] [N,HWA], deltas[l] [N,HWA,4]; labels [N,A] in {-1,0,1}; matched_gt [N,A,4]."""
    N = labels.shape[0]
    lg = torch.cat(logits, 1)
    dl = torch.cat(deltas, 1)
    pos = labels == 1
    tgt = torch.stack([get_deltas(anchors, matched_gt[n], cfg.rpn_weights) for n in range(N)])
    loc = (dl[pos] - tgt[pos]).abs().sum()  # smooth_l1 with beta = 0
    valid = labels >= 0
    obj = F.binary_cross_entropy_with_logits(lg[valid], labels[valid].float(), reduction="sum")
    norm = tc.rpn_batch_per_image * N
    return {"loss_rpn_cls": obj / norm, "loss_rpn_loc": loc / norm}


GT_LOGIT = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))  # add_ground_truth_to_proposals


def match_proposals(prop_boxes: torch.Tensor, gt_boxes: torch.Tensor, gt_classes: torch.Tensor, cfg: O.OracleCfg, tc: TrainCfg):
    """ROIHeads.label_and_sample_proposals up to the sampling: appends gt boxes, matches at IoU 0.5.
    -> boxes [R+G,4], class label per box (num_classes = background) , matched gt index."""
    boxes = torch.cat([prop_boxes, gt_boxes], 0) if tc.proposal_append_gt else prop_boxes
    midx, mlab = matcher(pairwise_iou(gt_boxes, boxes), (tc.roi_iou_threshold,), (0, 1), False)
    if len(gt_boxes):
        cls = gt_classes[midx].clone()
        cls[mlab == 0] = cfg.num_classes
        cls[mlab == -1] = -1
    else:
        cls = torch.full_like(midx, cfg.num_classes)
    return boxes, cls, midx


def sample_proposals(cls: torch.Tensor, cfg: O.OracleCfg, tc: TrainCfg, gen) -> torch.Tensor:
    fg, bg = subsample_labels(cls, tc.roi_batch_per_image, tc.roi_positive_fraction, cfg.num_classes, gen)
    return torch.cat([fg, bg], 0)


def box_losses(scores: torch.Tensor, deltas: torch.Tensor, prop_boxes, gt_classes, gt_boxes, cfg: O.OracleCfg):
    """FastRCNNOutputLayers.losses: mean cross entropy + L1 on the gt class's deltas of foreground rows / all rows."""
    loss_cls = F.cross_entropy(scores, gt_classes, reduction="mean")
    fg = ((gt_classes >= 0) & (gt_classes < cfg.num_classes)).nonzero().squeeze(1)
    pred = deltas.view(-1, cfg.num_classes, 4)[fg, gt_classes[fg]]
    tgt = get_deltas(prop_boxes[fg], gt_boxes[fg], cfg.box_weights)
    loss_box = (pred - tgt).abs().sum() / max(gt_classes.numel(), 1.0)
    return {"loss_cls": loss_cls, "loss_box_reg": loss_box}


# ----------------------------------------------------------------------------------------------
# differentiable ROIPooler (C forward / C backward)
# ----------------------------------------------------------------------------------------------
class _RoiAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, P, scale, ratio, aligned):
        ctx.save_for_backward(rois)
        ctx.args = (tuple(feat.shape), P, scale, ratio, aligned)
        return O.roi_align(feat.detach(), rois, P, scale, ratio, aligned)

    @staticmethod
    def backward(ctx, dout):
        (rois,) = ctx.saved_tensors
        (N, C, H, W), P, scale, ratio, aligned = ctx.args
        dfeat = torch.zeros(N, C, H, W, dtype=torch.float64)
        dout = dout.contiguous().float()
        rois = rois.contiguous().float()
        if rois.shape[0]:
            O._lib().orc_roi_align_backward_nchw(
                ctypes.c_void_p(dout.data_ptr()), N, C, H, W, ctypes.c_void_p(rois.data_ptr()), rois.shape[0], P,
                ctypes.c_float(scale), int(ratio), int(bool(aligned)), ctypes.c_void_p(dfeat.data_ptr()))
        return dfeat.float(), None, None, None, None, None


def roi_pool_fpn_diff(feats, box_lists, P, ratio, aligned):
    names = ("p2", "p3", "p4", "p5")
    rois = torch.cat([torch.cat((torch.full((len(b), 1), float(i)), b.float()), 1) for i, b in enumerate(box_lists)], 0)
    lv = O.assign_levels(rois[:, 1:])
    out = torch.zeros(rois.shape[0], feats["p2"].shape[1], P, P)
    for li, name in enumerate(names):
        sel = (lv == li).nonzero().squeeze(1)
        if len(sel):
            out = out.index_copy(0, sel, _RoiAlignFn.apply(feats[name], rois[sel], P, 1.0 / O.FPN_STRIDES[name], ratio, aligned))
    return out


# ----------------------------------------------------------------------------------------------
# the step
# ----------------------------------------------------------------------------------------------
def synthetic_targets(n: int, seed: int = 2020, h: int = 480, w: int = 640):
    """Random ground truth for the synthetic training batch: 2-6 boxes per image, classes in {0,1}."""
    rng = np.random.default_rng(seed + 7)
    out = []
    for _ in range(n):
        g = int(rng.integers(2, 7))
        bw, bh = rng.uniform(40, 320, g), rng.uniform(40, 260, g)
        x1, y1 = rng.uniform(0, w - bw), rng.uniform(0, h - bh)
        boxes = np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
        out.append((torch.from_numpy(boxes), torch.from_numpy(rng.integers(0, 2, g).astype(np.int64))))
    return out


def forward_losses(images_chw: List[torch.Tensor], targets, P: Dict[str, torch.Tensor], cfg: O.OracleCfg, tc: TrainCfg,
                   gen: Optional[torch.Generator] = None, samples: Optional[dict] = None):
    """Training forward (planercnn.py:83-123).  `samples` (optional) fixes the random draws:
    {"anchor_labels": [N,A] int8 after sampling, "roi_idx": [per image LongTensor of sampled rows]}.
    Returns (losses dict, aux dict with every intermediate the parity tests compare)."""
    x, sizes = O.preprocess(images_chw, cfg)
    feats = O.backbone(x, P)
    logits, deltas = O.rpn_head(feats, P)
    names = ("p2", "p3", "p4", "p5", "p6")
    feat_hw = [tuple(feats[n].shape[-2:]) for n in names]
    anchors = all_anchors(feat_hw, cfg)
    gt_boxes = [t[0] for t in targets]
    gt_classes = [t[1] for t in targets]
    N = len(images_chw)
    matched = match_anchors(anchors, gt_boxes, tc)
    if samples is None:
        labels = torch.stack([sample_anchors(m[1], tc, gen) for m in matched])
    else:
        labels = samples["anchor_labels"]
    matched_gt = torch.stack([gb[m[0]] if len(gb) else torch.zeros_like(anchors) for gb, m in zip(gt_boxes, matched)])
    losses = rpn_losses(logits, deltas, anchors, labels, matched_gt, cfg, tc)

    with torch.no_grad():  # proposals carry no gradient (d2 RPN.predict_proposals)
        pcfg = O.OracleCfg(**{**cfg.__dict__, "rpn_pre_topk": tc.rpn_pre_topk_train, "rpn_post_topk": tc.rpn_post_topk_train})
        props = O.rpn_select([l.detach() for l in logits], [d.detach() for d in deltas], feat_hw, sizes, pcfg)
        if samples is not None and "proposals" in samples:  # stage-wise parity: proposal selection is discontinuous in
            props = [(b, None) for b in samples["proposals"]]  # its inputs, so the test feeds the HIP path's own boxes
    sel_boxes, sel_cls, sel_gt, roi_idx, match_cls = [], [], [], [], []
    for n in range(N):
        boxes, cls, midx = match_proposals(props[n][0], gt_boxes[n], gt_classes[n], cfg, tc)
        idx = sample_proposals(cls, cfg, tc, gen) if samples is None else samples["roi_idx"][n]
        roi_idx.append(idx)
        match_cls.append(cls)
        sel_boxes.append(boxes[idx])
        sel_cls.append(cls[idx])
        sel_gt.append(gt_boxes[n][midx[idx]] if len(gt_boxes[n]) else boxes[idx])
    pooled = roi_pool_fpn_diff(feats, sel_boxes, *cfg.box_pool)
    scores, bdeltas = O.box_predictor(O.box_head(pooled, P), P)
    losses.update(box_losses(scores, bdeltas, torch.cat(sel_boxes), torch.cat(sel_cls), torch.cat(sel_gt), cfg))
    aux = dict(feats=feats, logits=logits, deltas=deltas, anchors=anchors, anchor_match=matched, anchor_labels=labels,
               matched_gt=matched_gt, proposals=props, roi_idx=roi_idx, roi_match_cls=match_cls, roi_boxes=sel_boxes,
               roi_cls=sel_cls, roi_gt=sel_gt, pooled=pooled, scores=scores, box_deltas=bdeltas)
    return losses, aux


def loss_and_grads(images_chw, targets, P, cfg: O.OracleCfg, tc: TrainCfg, gen=None, samples=None):
    """-> (losses, grads {name: tensor} for trainable_names(P), aux)."""
    names = trainable_names(P)
    Pg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in P.items()}
    losses, aux = forward_losses(images_chw, targets, Pg, cfg, tc, gen, samples)
    total = sum(losses.values())
    total.backward()
    grads = {k: (Pg[k].grad if Pg[k].grad is not None else torch.zeros_like(Pg[k])) for k in names}
    return {k: v.detach() for k, v in losses.items()}, grads, aux


def lr_at(it: int, tc: TrainCfg) -> float:
    """WarmupMultiStepLR (linear warm-up)."""
    f = 1.0
    if it < tc.warmup_iters:
        a = it / tc.warmup_iters
        f = tc.warmup_factor * (1 - a) + a
    return tc.base_lr * f * tc.gamma ** sum(1 for s in tc.steps if s <= it)


def sgd_step(P, grads, bufs: Dict[str, torch.Tensor], lr: float, tc: TrainCfg):
    """torch.optim.SGD (momentum, weight decay, no dampening / nesterov), in place on P."""
    for k, g in grads.items():
        d = g + tc.weight_decay * P[k]
        if k not in bufs:
            bufs[k] = d.clone()
        else:
            bufs[k].mul_(tc.momentum).add_(d)
        P[k] = P[k] - lr * bufs[k]
