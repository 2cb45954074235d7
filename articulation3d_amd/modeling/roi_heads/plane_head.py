"""Plane-normal head, registry name `PlaneRCNNConvFCHead` in ROI_PLANE_HEAD_REGISTRY.

Follows pkg/modeling/roi_heads/plane_head.py: 4x[conv3x3 256->256 + bias + ReLU] (:41-53), flatten,
FC 50176->1024 + ReLU (:57,78), FC 1024->3 (:62,80), L2-normalise (:81-82), `plane_rcnn_inference`
(:127-132).  Parameter names as in the reference (`plane_conv{k}`, `plane_fc{k}`, `param_pred`)."""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from ... import ops
from ...registry import ROI_PLANE_HEAD_REGISTRY
from ...structures import ShapeSpec
from ..layers import ACT_RELU, Conv2d, Linear, c2_msra_fill, c2_xavier_fill, to_nhwc


def head_fc(x_rows: torch.Tensor, fc: Linear) -> torch.Tensor:
    """[rows, K] x [N, K]^T with K = 50176: weight-bandwidth bound at small row counts -> split-K."""
    M, K = x_rows.shape
    p = fc.packed()
    return ops.linear(x_rows, p, splitk=ops.choose_splitk(M, p.cols, K))


@ROI_PLANE_HEAD_REGISTRY.register()
class PlaneRCNNConvFCHead(nn.Module):
    def __init__(self, cfg, input_shape: ShapeSpec):
        super().__init__()
        h = cfg.MODEL.ROI_PLANE_HEAD
        assert h.NORM == "", "reference config: ROI_PLANE_HEAD.NORM ''"
        self._plane_normal_only = h.NORMAL_ONLY
        self._output_size = (input_shape.channels, input_shape.height, input_shape.width)
        self.conv_norm_relus = []
        for k in range(h.NUM_CONV):
            conv = Conv2d(self._output_size[0], h.CONV_DIM, 3, padding=1, act=ACT_RELU)
            self.add_module(f"plane_conv{k + 1}", conv)
            self.conv_norm_relus.append(conv)
            self._output_size = (h.CONV_DIM, self._output_size[1], self._output_size[2])
        self.fcs = []
        for k in range(h.NUM_FC):
            chw = self._output_size if isinstance(self._output_size, tuple) else None
            fc = Linear(int(np.prod(self._output_size)), h.FC_DIM, chw=chw, act=ACT_RELU)
            self.add_module(f"plane_fc{k + 1}", fc)
            self.fcs.append(fc)
            self._output_size = h.FC_DIM
        self.param_pred = Linear(h.FC_DIM, h.PARAM_DIM)
        for layer in self.conv_norm_relus:
            c2_msra_fill(layer.weight, layer.bias)
        for layer in self.fcs:
            c2_xavier_fill(layer.weight, layer.bias)
        self._loss_weight = h.LOSS_WEIGHT
        self.keep_raw, self.raw = False, None

    @property
    def output_size(self):
        return self._output_size

    def forward_rows(self, x):
        """x: [rows,14,14,C] NHWC pooled features -> [rows, 3] unit normals."""
        for layer in self.conv_norm_relus:
            x = layer(x, wino=True)  # fixed algorithm choice: the ROI count must not change a ROI's result
        x = ops.keep_amax(x.reshape(x.shape[0], -1), x)  # (a view: the recorded per-ROI maxima stay valid)
        for fc in self.fcs:
            x = head_fc(x, fc)
        n = self.param_pred.out_features
        if self.keep_raw:  # checker hook (tests, bench): the param_pred output before F.normalize (plane_head.py:80), same kernel
            self.raw = ops.linear_small(x, self.param_pred.weight, self.param_pred.bias, norm_n=0)
        return ops.linear_small(x, self.param_pred.weight, self.param_pred.bias,
                                norm_n=n if self._plane_normal_only else 0)

    def forward(self, x, instances):
        """Reference signature (plane_head.py:71-89): x = pooled NCHW features; mutates and returns instances."""
        if self.training:
            raise NotImplementedError("plane loss (training) is outside the inference hot path (SURVEY.md 8f-1)")
        planes = self.forward_rows(to_nhwc(x)) if x.shape[0] else x.new_zeros((0, self.param_pred.out_features))
        plane_rcnn_inference(planes, instances)
        return instances


def plane_rcnn_inference(plane_pred, pred_instances):
    num_boxes_per_image = [len(i) for i in pred_instances]
    for plane, instances in zip(plane_pred.split(num_boxes_per_image, dim=0), pred_instances):
        instances.pred_plane = plane


def build_plane_head(cfg, input_shape):
    return ROI_PLANE_HEAD_REGISTRY.get(cfg.MODEL.ROI_PLANE_HEAD.NAME)(cfg, input_shape)
