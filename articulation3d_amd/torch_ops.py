"""`torch.ops.a3d.*`: the HIP kernels as registered PyTorch custom operators (BASELINE.json north_star: "surfaced to Python
via PyTorch-ROCm custom ops").

The reference reaches its native operators through registered torch ops -- `torchvision.ops.roi_align` / `nms` /
`batched_nms` behind detectron2's ROIPooler, RPN and FastRCNNOutputLayers (pkg/modeling/roi_heads/roi_heads.py:50-55,185,
236,250,268), `F.conv2d` / `F.linear` for the layers, `F.grid_sample` in the mask paste (pkg/layers/mask_ops.py:60).  The
replacements are registered the same way, so a maintainer swaps them in at those call sites and
dispatcher-level tooling sees them as ops.  Schemas are defined with `torch.library`; the implementation registered for the
CUDA (= HIP on ROCm) dispatch key hands `data_ptr()`s to the C ABI of include/a3d.h through `articulation3d_amd.ops`.
There is deliberately NO CPU implementation: calling an op with CPU tensors fails in the dispatcher
("Could not run 'a3d::...' with arguments from the 'CPU' backend").

    torch.ops.a3d.roi_align_fpn(feats, scales, boxes, count, P, sampling_ratio, aligned)       -> [B*R, P, P, C]
    torch.ops.a3d.group_nms(g_boxes, g_valid, g_n, thresh)                                     -> keep [G, 1024] int32
    torch.ops.a3d.rpn_proposals(heads, strides, cell_anchors, img_h, img_w, pre, post, thr, min_size, weights, clamp)
                                                                                           -> boxes, logits, level, pos, count
    torch.ops.a3d.box_detections(pred, prop_boxes, prop_count, img_h, img_w, classes, score_thr, nms_thr, topk, weights, clamp)
                                                                                           -> boxes, scores, classes, pos, count
    torch.ops.a3d.conv2d_fused(x, w, scale, shift, res, w_wino, KH, KW, stride, pad, act)      -> y (NHWC)
    torch.ops.a3d.paste_lsq(boxes, scores, count, row_offset, mask_prob, normals, depth, H, W, post_thr, mask_thr, want_masks)
                                                                                           -> masks, planes, area, keep, boxes
    torch.ops.a3d.detections_pack(boxes, scores, classes, count, row_offset, keep, planes, rot_axis, tran_axis, mask_prob)
                                                                                           -> records, rec_count
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import ops

_lib = torch.library.Library("a3d", "DEF")

_lib.define("roi_align_fpn(Tensor[] feats, float[] scales, Tensor boxes, Tensor? count, int P, int sampling_ratio, bool aligned) -> Tensor")
_lib.define("group_nms(Tensor g_boxes, Tensor g_valid, Tensor g_n, float thresh) -> Tensor")
_lib.define("rpn_proposals(Tensor[] heads, int[] strides, Tensor cell_anchors, int img_h, int img_w, int pre_topk, int post_topk, "
            "float nms_thresh, float min_size, float[] weights, float scale_clamp) -> (Tensor, Tensor, Tensor, Tensor, Tensor)")
_lib.define("box_detections(Tensor pred, Tensor prop_boxes, Tensor prop_count, int img_h, int img_w, int num_classes, float score_thresh, "
            "float nms_thresh, int topk, float[] weights, float scale_clamp) -> (Tensor, Tensor, Tensor, Tensor, Tensor)")
_lib.define("conv2d_fused(Tensor x, Tensor w, Tensor? scale, Tensor? shift, Tensor? res, Tensor? w_wino, int KH, int KW, int stride, int pad, "
            "int act) -> Tensor")
_lib.define("paste_lsq(Tensor boxes, Tensor scores, Tensor count, Tensor row_offset, Tensor mask_prob, Tensor? normals, Tensor? depth, "
            "int img_h, int img_w, float post_score_thresh, float mask_thresh, bool want_masks) -> (Tensor, Tensor, Tensor, Tensor, Tensor)")
_lib.define("detections_pack(Tensor boxes, Tensor scores, Tensor classes, Tensor count, Tensor row_offset, Tensor keep, Tensor planes, "
            "Tensor rot_axis, Tensor tran_axis, Tensor mask_prob) -> (Tensor, Tensor)")


def _roi_align_fpn(feats: List[Tensor], scales: List[float], boxes: Tensor, count: Optional[Tensor], P: int, sampling_ratio: int,
                   aligned: bool) -> Tensor:
    return ops.roi_align_fpn(list(feats), list(scales), boxes, count, P, sampling_ratio, aligned)


def _group_nms(g_boxes: Tensor, g_valid: Tensor, g_n: Tensor, thresh: float) -> Tensor:
    return ops.group_nms(g_boxes, g_valid, g_n, thresh)


def _rpn_proposals(heads: List[Tensor], strides: List[int], cell_anchors: Tensor, img_h: int, img_w: int, pre_topk: int, post_topk: int,
                   nms_thresh: float, min_size: float, weights: List[float], scale_clamp: float):
    return ops.rpn_proposals(list(heads), list(strides), cell_anchors.cpu(), (img_h, img_w), pre_topk=pre_topk, post_topk=post_topk,
                             nms_thresh=nms_thresh, min_size=min_size, weights=tuple(weights), scale_clamp=scale_clamp)


def _box_detections(pred: Tensor, prop_boxes: Tensor, prop_count: Tensor, img_h: int, img_w: int, num_classes: int, score_thresh: float,
                    nms_thresh: float, topk: int, weights: List[float], scale_clamp: float):
    return ops.box_detections(pred, prop_boxes, prop_count, (img_h, img_w), num_classes=num_classes, score_thresh=score_thresh,
                              nms_thresh=nms_thresh, topk=topk, weights=tuple(weights), scale_clamp=scale_clamp)


def _conv2d_fused(x: Tensor, w: Tensor, scale: Optional[Tensor], shift: Optional[Tensor], res: Optional[Tensor], w_wino: Optional[Tensor],
                  KH: int, KW: int, stride: int, pad: int, act: int) -> Tensor:
    return ops.conv2d(x, _packed_for(w, scale, shift, w_wino, KH, KW, stride, pad, act), res=res)


_PACKED: "dict" = {}  # opt-in cache: (identity of the filter tensors, geometry) -> PackedConv with its cached splits / scales
_CACHE_FILTERS = False


def enable_filter_cache(flag: bool = True) -> None:
    """OPT-IN cache of the derived filter forms of `conv2d_fused` (power-of-two scale, fp16 / bf16 planes).  Off by default: the op
    takes RAW tensors, and nothing the op can see cheaply tells it that a filter's bytes changed -- `Tensor._version` is not bumped by
    writes through `.data`, by raw-pointer kernels (this package's own `a3d_sgd_momentum` on flat-buffer views) or by other libraries,
    so a cached split could silently go stale.  A caller that enables it owns the contract: call `invalidate_filter_cache()` after
    every update of a filter that does not go through an in-place torch op on that very tensor."""
    global _CACHE_FILTERS
    _CACHE_FILTERS = bool(flag)
    if not flag:
        _PACKED.clear()


def invalidate_filter_cache() -> None:
    _PACKED.clear()


def _packed_for(w, scale, shift, w_wino, KH, KW, stride, pad, act):
    """The op takes raw tensors, the kernels want the filter's derived forms.  By default they are built PER CALL from the bytes the
    tensors hold now (`presplit=False`: the direct kernels split the filter in their loaders, the Winograd layers split U on the launch
    stream, the power-of-two scale is read back once per call) -- correct for any way the caller updates its weights.  The detector's
    own modules do not come through here: they keep packed weights keyed on their parameters (modeling/layers.py)."""
    cols, Kpad = w.shape
    if not _CACHE_FILTERS:
        return ops.PackedConv(w, scale, shift, KH, KW, stride, pad, Kpad // (KH * KW), cols, Kpad, act, w_wino=w_wino, presplit=False)
    ident = lambda t: None if t is None else (t.data_ptr(), t._version, tuple(t.shape), t.device)
    key = (ident(w), ident(scale), ident(shift), ident(w_wino), KH, KW, stride, pad, act)
    p = _PACKED.pop(key, None)
    if p is None:
        p = ops.PackedConv(w, scale, shift, KH, KW, stride, pad, Kpad // (KH * KW), cols, Kpad, act, w_wino=w_wino, presplit=True)
        while len(_PACKED) >= 256:
            _PACKED.pop(next(iter(_PACKED)))
    _PACKED[key] = p  # (re-inserted last: most recently used)
    return p


def _paste_lsq(boxes: Tensor, scores: Tensor, count: Tensor, row_offset: Tensor, mask_prob: Tensor, normals: Optional[Tensor],
               depth: Optional[Tensor], img_h: int, img_w: int, post_score_thresh: float, mask_thresh: float, want_masks: bool):
    masks, planes, area, keep, out_boxes = ops.paste_lsq(boxes, scores, count, row_offset, mask_prob, normals, depth, (img_h, img_w),
                                                        post_score_thresh=post_score_thresh, mask_thresh=mask_thresh, want_masks=want_masks)
    if masks is None:  # an op returns tensors: an empty one stands for "not materialised"
        masks = torch.empty((0,), device=boxes.device, dtype=torch.uint8)
    return masks, planes, area, keep, out_boxes


def _detections_pack(boxes: Tensor, scores: Tensor, classes: Tensor, count: Tensor, row_offset: Tensor, keep: Tensor, planes: Tensor,
                     rot_axis: Tensor, tran_axis: Tensor, mask_prob: Tensor) -> Tuple[Tensor, Tensor]:
    return ops.detections_pack(boxes, scores, classes, count, row_offset, keep, planes, rot_axis, tran_axis, mask_prob, mask_prob.shape[-1])


for _name, _fn in (("roi_align_fpn", _roi_align_fpn), ("group_nms", _group_nms), ("rpn_proposals", _rpn_proposals),
                   ("box_detections", _box_detections), ("conv2d_fused", _conv2d_fused), ("paste_lsq", _paste_lsq),
                   ("detections_pack", _detections_pack)):
    _lib.impl(_name, _fn, "CUDA")

OPS = ("roi_align_fpn", "group_nms", "rpn_proposals", "box_detections", "conv2d_fused", "paste_lsq", "detections_pack")
