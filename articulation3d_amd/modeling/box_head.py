"""Fast R-CNN box head + predictor (`FastRCNNConvFCHead`, `FastRCNNOutputLayers`), SURVEY.md A.8.

Replaces the detectron2 modules the reference reaches at pkg/modeling/roi_heads/roi_heads.py:186-187,206.
fc1/fc2 are MFMA GEMMs over all B*R proposals at once; cls_score + bbox_pred are one fused 1024->11 GEMM;
softmax, per-class decode, threshold, per-class NMS and top-k happen in the selection kernels."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from ..registry import ROI_BOX_HEAD_REGISTRY
from ..structures import ShapeSpec
from .layers import ACT_RELU, Linear, _Packable, c2_xavier_fill
from .rpn import SCALE_CLAMP


@ROI_BOX_HEAD_REGISTRY.register()
class FastRCNNConvFCHead(nn.Module):
    def __init__(self, cfg, input_shape: ShapeSpec):
        super().__init__()
        h = cfg.MODEL.ROI_BOX_HEAD
        assert h.NUM_CONV == 0 and h.NORM == "", "reference config: box head has no convs / norm"
        c, hh, ww = input_shape.channels, input_shape.height, input_shape.width
        self.fcs = []
        dim = c * hh * ww
        chw = (c, hh, ww)
        for k in range(h.NUM_FC):
            fc = Linear(dim, h.FC_DIM, chw=chw, act=ACT_RELU)
            c2_xavier_fill(fc.weight, fc.bias)
            self.add_module(f"fc{k + 1}", fc)
            self.fcs.append(fc)
            dim, chw = h.FC_DIM, None
        self._output_size = dim

    @property
    def output_shape(self):
        return ShapeSpec(channels=self._output_size)

    def forward(self, x):
        """x: [rows, P, P, C] NHWC bins (or [rows, K]) -> [rows, FC_DIM]."""
        if x.dtype != torch.float16:  # (pre-split rows from the pooler go to fc1 as they are: ops.linear)
            x = ops.keep_amax(x.reshape(x.shape[0], -1), x)  # (a view: the recorded per-ROI maxima stay valid)
        for fc in self.fcs:
            x = fc(x)
        return x


class _FusedPredictor(_Packable):
    def __init__(self, owner):
        super().__init__()
        object.__setattr__(self, "_o", owner)

    def _key(self):
        o = self._o
        ts = [o.cls_score.weight, o.cls_score.bias, o.bbox_pred.weight, o.bbox_pred.bias]
        return tuple((t.data_ptr(), t._version, str(t.device)) for t in ts)

    def _pack(self):
        o = self._o
        return ops.pack_fused_rows([o.cls_score.weight, o.bbox_pred.weight], [o.cls_score.bias, o.bbox_pred.bias],
                                   device=o.cls_score.weight.device)


class FastRCNNOutputLayers(nn.Module):
    def __init__(self, cfg, input_shape: ShapeSpec):
        super().__init__()
        dim = input_shape.channels
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        assert not cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG, "reference config: class-specific box regression"
        self.cls_score = Linear(dim, self.num_classes + 1)
        self.bbox_pred = Linear(dim, self.num_classes * 4)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        nn.init.constant_(self.cls_score.bias, 0)
        nn.init.constant_(self.bbox_pred.bias, 0)
        self.box_weights = tuple(cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS)
        self.test_score_thresh = cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST
        self.test_nms_thresh = cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST
        self.test_topk_per_image = cfg.TEST.DETECTIONS_PER_IMAGE
        self._fused = _FusedPredictor(self)

    def forward(self, x):
        """[rows, dim] -> [rows, 12]: class logits 0..2, per-class deltas 3..10, pad."""
        return ops.linear(x, self._fused.packed())

    def inference_batched(self, pred, prop_boxes, prop_count, img_hw, return_groups=False):
        return ops.box_detections(pred, prop_boxes, prop_count, img_hw, num_classes=self.num_classes,
                                  score_thresh=self.test_score_thresh, nms_thresh=self.test_nms_thresh,
                                  topk=self.test_topk_per_image, weights=self.box_weights, scale_clamp=SCALE_CLAMP,
                                  return_groups=return_groups)


def build_box_head(cfg, input_shape):
    return ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, input_shape)
