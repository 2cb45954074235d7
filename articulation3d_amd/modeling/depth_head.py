"""`PlaneRCNNDepthHead` (DEPTH_HEAD_REGISTRY): monocular depth from the pyramid.

Follows pkg/modeling/depth_net/depth_head.py: conv2d / deconv2d blocks (:32-46), layer list (:58-68),
forward (:72-89): p6 -> conv1 -> deconv1 -> bilinear to p5's size -> cat with conv2(p5) -> deconv2 -> ... ->
deconv5 (64 ch @ 240x320) -> depth_pred 64->1 -> bilinear x2.  Parameter names are those of the reference's
nn.Sequential blocks (`conv1.0.weight`, `conv1.1.running_mean`, `deconv1.1.weight`, `deconv1.2.*`, ...).

MI355X mapping: every 3x3 conv is one MFMA implicit-GEMM launch with BatchNorm (eval, eps 1e-3) folded
into scale/shift; the nearest x2 upsampling and the channel concat are done by the kernel's gather (neither
tensor is materialised); the 64->1 prediction conv is a wave-reduction kernel."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from ..registry import DEPTH_HEAD_REGISTRY
from .layers import ACT_LEAKY, ACT_RELU, BNConv2d, _CalibrationState, to_nhwc

__all__ = ["build_depth_head", "PlaneRCNNDepthHead", "DEPTH_HEAD_REGISTRY"]


def conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=None):
    return nn.Sequential(
        nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding),
        nn.BatchNorm2d(out_channels, eps=0.001, momentum=0.01),
        nn.LeakyReLU(inplace=True),
    )


def deconv2d(scale_factor=2, mode="nearest", in_channels=256, out_channels=128, kernel_size=3, stride=1, padding=1):
    return nn.Sequential(
        nn.Upsample(scale_factor=scale_factor, mode=mode),
        nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding),
        nn.BatchNorm2d(out_channels, eps=0.001, momentum=0.01),
        nn.ReLU(inplace=True),
    )


@DEPTH_HEAD_REGISTRY.register()
class PlaneRCNNDepthHead(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        for i in range(1, 6):
            setattr(self, f"conv{i}", conv2d(256, 128, 3, 1, 1))
        self.deconv1 = deconv2d(in_channels=128, out_channels=128)
        self.deconv2 = deconv2d(in_channels=256, out_channels=128)
        self.deconv3 = deconv2d(in_channels=256, out_channels=128)
        self.deconv4 = deconv2d(in_channels=256, out_channels=128)
        self.deconv5 = deconv2d(in_channels=256, out_channels=64)
        self.depth_pred = nn.Conv2d(64, 1, kernel_size=3, stride=1, padding=1)
        self._loss_weight = cfg.MODEL.DEPTH_HEAD.LOSS_WEIGHT
        self._freeze = cfg.MODEL.FREEZE
        # packers are plain attributes (not sub-modules) so the state_dict keeps the reference's names only
        object.__setattr__(self, "_pk_conv", [BNConv2d(getattr(self, f"conv{i}")[0], getattr(self, f"conv{i}")[1], ACT_LEAKY) for i in range(1, 6)])
        object.__setattr__(self, "_pk_deconv", [BNConv2d(getattr(self, f"deconv{i}")[1], getattr(self, f"deconv{i}")[2], ACT_RELU) for i in range(1, 6)])
        self._pred_w = None
        self._pred_key = None

    def _pred_packed(self):
        w = self.depth_pred.weight
        key = (w.data_ptr(), w._version, self.depth_pred.bias._version)
        if self._pred_w is None or key != self._pred_key:
            self._pred_w = (w.detach().float()[0].permute(1, 2, 0).contiguous(), float(self.depth_pred.bias.detach().float().item()))
            self._pred_key = key
        return self._pred_w

    def forward_nhwc(self, feats):
        """feats: {p2..p6} NHWC -> depth [B, 2*H_p2*2, 2*W_p2*2] (480x640 for 480x640 frames)."""
        C = lambda i, x: self._pk_conv[i - 1](x)
        D = lambda i, a, b=None: self._pk_deconv[i - 1](a, x2=b, ups=True)
        x = D(1, C(1, feats["p6"]))
        p5 = feats["p5"]
        x = ops.resize_bilinear(x, p5.shape[1], p5.shape[2])  # depth_head.py:82
        x = D(2, C(2, p5), x)
        x = D(3, C(3, feats["p4"]), x)
        x = D(4, C(4, feats["p3"]), x)
        w, b = self._pred_packed()
        c5 = C(5, feats["p2"])
        # deconv5 + depth_pred without the 64-channel 240x320 tensor between them (default arithmetic; ops.conv2d_ups_to1)
        # (not while batch-norm statistics are being calibrated: deconv5's BN is calibrated in BNConv2d.forward, which the fused form skips)
        fused = not self.training and not _CalibrationState.active
        d = ops.conv2d_ups_to1(c5, self._pk_deconv[4].packed_phases(), w, b, x2=x) if fused else None
        if d is None:
            x = D(5, c5, x)  # [B,240,320,64]
            d = ops.conv3x3_to1(x, w, b)  # [B,240,320]
        B, H, W = d.shape
        return ops.resize_bilinear(d.view(B, H, W, 1), 2 * H, 2 * W).view(B, 2 * H, 2 * W)  # depth_head.py:88-89

    def forward(self, features, gt_depth=None):
        """Reference signature (planercnn.py:174): features dict (NCHW-shaped) -> pred_depth [N,480,640]."""
        if self.training:
            raise NotImplementedError("depth loss (training) is outside the inference hot path (SURVEY.md 8f-1)")
        return self.forward_nhwc({k: to_nhwc(v) for k, v in features.items()})


def build_depth_head(cfg):
    return DEPTH_HEAD_REGISTRY.get(cfg.MODEL.DEPTH_HEAD.NAME)(cfg)
