"""RPN proposal generator (`RPN`, `StandardRPNHead`, `DefaultAnchorGenerator`), SURVEY.md A.4-A.6.

Replaces the detectron2 proposal generator the reference builds at planercnn.py:30 and calls at :168.
The 3x3 conv runs per level on the MFMA kernel; objectness + deltas are ONE fused 256->15 GEMM; anchors
are never materialised: the selection kernel regenerates the anchor of each surviving index.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
from torch import nn

from .. import ops
from ..streams import SMALL_BATCH, run_branches
from ..registry import ANCHOR_GENERATOR_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, RPN_HEAD_REGISTRY
from ..structures import Boxes, ImageList, Instances
from .layers import ACT_RELU, Conv2d, _Packable, to_nhwc

SCALE_CLAMP = math.log(1000.0 / 16)


@ANCHOR_GENERATOR_REGISTRY.register()
class DefaultAnchorGenerator(nn.Module):
    def __init__(self, cfg, input_shape):
        super().__init__()
        a = cfg.MODEL.ANCHOR_GENERATOR
        self.strides = [s.stride for s in input_shape]
        n = len(self.strides)
        sizes = list(a.SIZES) * n if len(a.SIZES) == 1 else list(a.SIZES)
        ratios = list(a.ASPECT_RATIOS) * n if len(a.ASPECT_RATIOS) == 1 else list(a.ASPECT_RATIOS)
        assert a.OFFSET == 0.0, "reference config: ANCHOR_GENERATOR.OFFSET 0"
        cells = []
        for s, r in zip(sizes, ratios):
            assert len(s) * len(r) == 3, "the selection kernel is built for 3 anchors per location (reference config)"
            cell = []
            for size in s:
                area = float(size) ** 2
                for ar in r:
                    w = math.sqrt(area / ar)
                    h = ar * w
                    cell.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
            cells.append(cell)
        self.cell_anchors = torch.tensor(cells, dtype=torch.float32)  # [L,3,4], python double -> fp32 as in d2
        self.num_cell_anchors = [3] * n

    @property
    def num_anchors(self):
        return self.num_cell_anchors


class _FusedRPNPredictors(_Packable):
    def __init__(self, head):
        super().__init__()
        object.__setattr__(self, "_head", head)

    def _key(self):
        h = self._head
        ts = [h.objectness_logits.weight, h.objectness_logits.bias, h.anchor_deltas.weight, h.anchor_deltas.bias]
        return tuple((t.data_ptr(), t._version, str(t.device)) for t in ts)

    def _pack(self):
        h = self._head
        return ops.pack_fused_rows([h.objectness_logits.weight, h.anchor_deltas.weight],
                                   [h.objectness_logits.bias, h.anchor_deltas.bias], device=h.conv.weight.device)


@RPN_HEAD_REGISTRY.register()
class StandardRPNHead(nn.Module):
    def __init__(self, cfg, input_shape):
        super().__init__()
        in_channels = input_shape[0].channels
        num_anchors, box_dim = 3, 4
        self.conv = Conv2d(in_channels, in_channels, 3, padding=1, act=ACT_RELU)
        self.objectness_logits = Conv2d(in_channels, num_anchors, 1)
        self.anchor_deltas = Conv2d(in_channels, num_anchors * box_dim, 1)
        for layer in (self.conv, self.objectness_logits, self.anchor_deltas):
            nn.init.normal_(layer.weight, std=0.01)
            nn.init.constant_(layer.bias, 0)
        self._fused = _FusedRPNPredictors(self)

    def forward_nhwc(self, feats: List[torch.Tensor]) -> List[torch.Tensor]:
        """-> per level [B,Hf,Wf,16]: channels 0..2 objectness, 3..14 deltas (a*4+coord), 15 unused."""
        fused = self._fused.packed()
        level = lambda f: ops.conv2d(self.conv(f), fused)
        if len(feats) >= 2 and feats[0].is_cuda and feats[0].shape[0] > SMALL_BATCH and not self.training:
            # the shared-filter 3x3 conv over ALL levels as one Winograd GEMM launch (ops.conv2d_levels: bit-identical to the per-level
            # launches, which it falls back to where the form does not apply)
            return [ops.conv2d(h, fused) for h in ops.conv2d_levels(feats, self.conv.packed())]
        if len(feats) >= 3 and feats[0].is_cuda and feats[0].shape[0] <= SMALL_BATCH:
            # 1-2 frames: the levels are independent chains of latency-bound launches -- three concurrent branches
            # (stride 4 | stride 8 | the rest), streams.py
            a, b, c = run_branches([lambda: [level(feats[0])], lambda: [level(feats[1])], lambda: [level(f) for f in feats[2:]]], True)
            return a + b + c
        return [level(f) for f in feats]


@PROPOSAL_GENERATOR_REGISTRY.register()
class RPN(nn.Module):
    def __init__(self, cfg, input_shape: Dict[str, "ShapeSpec"]):
        super().__init__()
        self.in_features = list(cfg.MODEL.RPN.IN_FEATURES)
        shapes = [input_shape[f] for f in self.in_features]
        self.anchor_generator = ANCHOR_GENERATOR_REGISTRY.get(cfg.MODEL.ANCHOR_GENERATOR.NAME)(cfg, shapes)
        self.rpn_head = RPN_HEAD_REGISTRY.get(cfg.MODEL.RPN.HEAD_NAME)(cfg, shapes)
        self.strides = [s.stride for s in shapes]
        self.pre_nms_topk = {True: cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, False: cfg.MODEL.RPN.PRE_NMS_TOPK_TEST}
        self.post_nms_topk = {True: cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN, False: cfg.MODEL.RPN.POST_NMS_TOPK_TEST}
        self.nms_thresh = cfg.MODEL.RPN.NMS_THRESH
        self.min_box_size = float(cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE)
        self.box_weights = tuple(cfg.MODEL.RPN.BBOX_REG_WEIGHTS)

    def forward_batched(self, feats_nhwc: Dict[str, torch.Tensor], img_hw, heads=None, return_groups=False):
        """Sync-free form: -> (boxes [B,K,4], logits [B,K], level, pos, count [B]) fixed-size device tensors."""
        if heads is None:
            heads = self.rpn_head.forward_nhwc([feats_nhwc[f] for f in self.in_features])
        return ops.rpn_proposals(
            heads, self.strides, self.anchor_generator.cell_anchors, img_hw,
            pre_topk=self.pre_nms_topk[False], post_topk=self.post_nms_topk[False], nms_thresh=self.nms_thresh,
            min_size=self.min_box_size, weights=self.box_weights, scale_clamp=SCALE_CLAMP, return_groups=return_groups)

    def forward(self, images: ImageList, features: Dict[str, torch.Tensor], gt_instances: Optional[list] = None):
        """Reference signature (planercnn.py:168): -> (list[Instances{proposal_boxes, objectness_logits}], {})."""
        if self.training:
            raise NotImplementedError("RPN losses (training) are outside the inference hot path (SURVEY.md 8f-1)")
        sizes = set(images.image_sizes)
        assert len(sizes) == 1, "batched frames must share one size (the reference feeds 480x640 frames)"
        hw = images.image_sizes[0]
        feats = {f: to_nhwc(features[f]) for f in self.in_features}
        boxes, logits, _lvl, _pos, count = self.forward_batched(feats, hw)
        out = []
        for b, n in enumerate(count.tolist()):
            inst = Instances(hw)
            inst.proposal_boxes = Boxes(boxes[b, :n])
            inst.objectness_logits = logits[b, :n]
            out.append(inst)
        return out, {}


def build_proposal_generator(cfg, input_shape):
    name = cfg.MODEL.PROPOSAL_GENERATOR.NAME
    if name == "PrecomputedProposals":
        return None
    return PROPOSAL_GENERATOR_REGISTRY.get(name)(cfg, input_shape)
