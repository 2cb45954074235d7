"""Articulation-axis head, registry name `PlaneRCNNConvFCHead` in ROI_AXIS_HEAD_REGISTRY.

Follows pkg/modeling/roi_heads/axis_head.py: two independent conv-FC towers R and T (:44-76);
rotation 1024->2 normalised || offset 1024->1 (:80-81,106-108) -> pred_rot_axis [sin, cos, offset];
translation 1024->2 normalised (:82,120) -> pred_tran_axis; `arti_inference` (:204-211).
Parameter names as in the reference (`axis_R_conv{k}`, `axis_T_fc{k}`, `rotation`, `offset`, `translation`)."""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from ... import ops
from ...registry import ROI_AXIS_HEAD_REGISTRY
from ...structures import ShapeSpec
from ..layers import ACT_RELU, Conv2d, Linear, c2_msra_fill, c2_xavier_fill, to_nhwc
from .plane_head import head_fc


@ROI_AXIS_HEAD_REGISTRY.register()
class PlaneRCNNConvFCHead(nn.Module):
    def __init__(self, cfg, input_shape: ShapeSpec):
        super().__init__()
        h = cfg.MODEL.ROI_AXIS_HEAD
        assert h.NORM == "", "reference config: ROI_AXIS_HEAD.NORM ''"
        self._output_size = (input_shape.channels, input_shape.height, input_shape.width)
        self.smooth_l1_beta = h.SMOOTH_L1_BETA
        self.conv_norm_relus_R, self.conv_norm_relus_T = [], []
        for k in range(h.NUM_CONV):
            for t, lst in (("R", self.conv_norm_relus_R), ("T", self.conv_norm_relus_T)):
                conv = Conv2d(self._output_size[0], h.CONV_DIM, 3, padding=1, act=ACT_RELU)
                self.add_module(f"axis_{t}_conv{k + 1}", conv)
                lst.append(conv)
            self._output_size = (h.CONV_DIM, self._output_size[1], self._output_size[2])
        self.fcs_R, self.fcs_T = [], []
        for k in range(h.NUM_FC):
            chw = self._output_size if isinstance(self._output_size, tuple) else None
            for t, lst in (("R", self.fcs_R), ("T", self.fcs_T)):
                fc = Linear(int(np.prod(self._output_size)), h.FC_DIM, chw=chw, act=ACT_RELU)
                self.add_module(f"axis_{t}_fc{k + 1}", fc)
                lst.append(fc)
            self._output_size = h.FC_DIM
        self.rotation = Linear(h.FC_DIM, 2)
        self.offset = Linear(h.FC_DIM, 1)
        self.translation = Linear(h.FC_DIM, 2)
        for layer in self.conv_norm_relus_R + self.conv_norm_relus_T:
            c2_msra_fill(layer.weight, layer.bias)
        for layer in self.fcs_R + self.fcs_T:
            c2_xavier_fill(layer.weight, layer.bias)
        self._loss_weight = h.LOSS_WEIGHT
        self.keep_raw, self.raw = False, None

    @property
    def output_size(self):
        return self._output_size

    def _tower(self, x, convs, fcs):
        for layer in convs:
            x = layer(x, wino=True)  # fixed algorithm choice: the ROI count must not change a ROI's result
        x = ops.keep_amax(x.reshape(x.shape[0], -1), x)  # (a view: the recorded per-ROI maxima stay valid)
        for fc in fcs:
            x = head_fc(x, fc)
        return x

    def forward_rows(self, x):
        """x: [rows,14,14,C] NHWC -> (pred_rot_axis [rows,3], pred_tran_axis [rows,2])."""
        xr = self._tower(x, self.conv_norm_relus_R, self.fcs_R)
        w = torch.cat((self.rotation.weight, self.offset.weight), 0).contiguous()
        b = torch.cat((self.rotation.bias, self.offset.bias), 0).contiguous()
        rot = ops.linear_small(xr, w, b, norm_n=2)  # normalise (sin, cos); offset passes through
        xt = self._tower(x, self.conv_norm_relus_T, self.fcs_T)
        tran = ops.linear_small(xt, self.translation.weight, self.translation.bias, norm_n=2)
        if self.keep_raw:  # checker hook: the rotation | offset and translation FC outputs before F.normalize (axis_head.py:106,120)
            self.raw = (ops.linear_small(xr, w, b, norm_n=0), ops.linear_small(xt, self.translation.weight, self.translation.bias, norm_n=0))
        return rot, tran

    def forward(self, x, instances):
        """Reference signature (axis_head.py:95-132)."""
        if self.training:
            raise NotImplementedError("axis loss (training) is outside the inference hot path (SURVEY.md 8f-1)")
        if x.shape[0]:
            rot, tran = self.forward_rows(to_nhwc(x))
        else:
            rot, tran = x.new_zeros((0, 3)), x.new_zeros((0, 2))
        arti_inference(rot, tran, instances)
        return instances


def arti_inference(pred_rot_axis, pred_tran_axis, pred_instances):
    n = [len(i) for i in pred_instances]
    for r_axis, t_axis, instances in zip(pred_rot_axis.split(n, dim=0), pred_tran_axis.split(n, dim=0), pred_instances):
        instances.pred_rot_axis = r_axis
        instances.pred_tran_axis = t_axis


def build_axis_head(cfg, input_shape):
    return ROI_AXIS_HEAD_REGISTRY.get(cfg.MODEL.ROI_AXIS_HEAD.NAME)(cfg, input_shape)
