"""Mask R-CNN head (`MaskRCNNConvUpsampleHead`), SURVEY.md A.9.  Replaces the detectron2 head reached at
pkg/modeling/roi_heads/roi_heads.py:237.  4x conv3x3+ReLU (MFMA), ConvTranspose2d k2 s2 as ONE GEMM with
a 2x2 pixel-shuffle store, then the class-agnostic 256->1 predictor fused with the sigmoid."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from ..registry import ROI_MASK_HEAD_REGISTRY
from ..structures import ShapeSpec
from .layers import ACT_RELU, Conv2d, _Packable, c2_msra_fill


class ConvTranspose2x2(_Packable):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cin, cout, 2, 2))
        self.bias = nn.Parameter(torch.zeros(cout))

    def _pack(self):
        return ops.pack_deconv2x2(self.weight, self.bias, ACT_RELU, device=self.weight.device)

    def forward(self, x, **kw):
        return ops.conv2d(x, self.packed(), **kw)


@ROI_MASK_HEAD_REGISTRY.register()
class MaskRCNNConvUpsampleHead(nn.Module):
    def __init__(self, cfg, input_shape: ShapeSpec):
        super().__init__()
        m = cfg.MODEL.ROI_MASK_HEAD
        assert m.NORM == "" and m.CLS_AGNOSTIC_MASK, "reference config: no norm, class-agnostic mask"
        cin, dim = input_shape.channels, m.CONV_DIM
        self.conv_norm_relus = []
        for k in range(m.NUM_CONV):
            conv = Conv2d(cin if k == 0 else dim, dim, 3, padding=1, act=ACT_RELU)
            c2_msra_fill(conv.weight, conv.bias)
            self.add_module(f"mask_fcn{k + 1}", conv)
            self.conv_norm_relus.append(conv)
        self.deconv = ConvTranspose2x2(dim if m.NUM_CONV > 0 else cin, dim)
        c2_msra_fill(self.deconv.weight, self.deconv.bias)
        self.predictor = Conv2d(dim, 1, 1)
        nn.init.normal_(self.predictor.weight, std=0.001)
        nn.init.constant_(self.predictor.bias, 0)

    def forward_rows(self, x):
        """x: [rows,14,14,C] NHWC -> mask probabilities [rows, 28, 28]."""
        rows = x.shape[0]
        for conv in self.conv_norm_relus:
            x = conv(x, wino=True)  # fixed algorithm choice: the ROI count must not change a ROI's result
        x = self.deconv(x)  # [rows,28,28,dim]
        S = x.shape[1]
        w = self.predictor.weight.reshape(1, -1).contiguous()
        p = ops.linear_small(x.view(rows * S * S, x.shape[3]), w, self.predictor.bias, sigmoid=True)
        return p.view(rows, S, S)


def build_mask_head(cfg, input_shape):
    return ROI_MASK_HEAD_REGISTRY.get(cfg.MODEL.ROI_MASK_HEAD.NAME)(cfg, input_shape)
