"""ResNet-50 + FPN backbone (`build_resnet_fpn_backbone`), SURVEY.md A.2-A.3.

Replaces the detectron2 backbone the reference builds at pkg/modeling/meta_arch/planercnn.py:29 and calls
at :150.  Parameter names follow detectron2 (`backbone.bottom_up.res2.0.conv1.weight`,
`backbone.fpn_lateral2.weight`, ...).  Every conv is one launch of the fp32-MFMA implicit-GEMM kernel
with FrozenBN / bias / ReLU / residual / top-down add fused in its epilogue.
"""
from __future__ import annotations

from typing import Dict

import torch
from torch import nn

from .. import ops
from ..streams import SMALL_BATCH, run_branches
from ..registry import BACKBONE_REGISTRY
from ..structures import ShapeSpec
from .layers import ACT_NONE, ACT_RELU, Conv2d, FrozenBatchNorm2d, c2_msra_fill, c2_xavier_fill, to_nchw_view


class BasicStem(nn.Module):
    def __init__(self, in_channels=3, out_channels=64):
        super().__init__()
        self.conv1 = Conv2d(in_channels, out_channels, 7, stride=2, padding=3, bias=False,
                            norm=FrozenBatchNorm2d(out_channels), act=ACT_RELU)
        c2_msra_fill(self.conv1.weight)

    def forward(self, x4):  # [B,H,W,4] normalised
        from .layers import _CalibrationState

        if not _CalibrationState.active and not self.training:
            y = ops.stem_pool(x4, self.conv1.packed())  # conv + BN + ReLU + max-pool in one launch (fp16x2 arithmetic; same bits)
            if y is not None:
                return y
        return ops.maxpool3x3s2(self.conv1(x4))


class BottleneckBlock(nn.Module):
    """1x1 (stride here: STRIDE_IN_1X1) -> 3x3 -> 1x1, + identity / projected shortcut, ReLU."""

    def __init__(self, in_channels, out_channels, bottleneck_channels, stride):
        super().__init__()
        self.shortcut = None
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, 1, stride=stride, bias=False,
                                   norm=FrozenBatchNorm2d(out_channels))
        self.conv1 = Conv2d(in_channels, bottleneck_channels, 1, stride=stride, bias=False,
                            norm=FrozenBatchNorm2d(bottleneck_channels), act=ACT_RELU)
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, 3, stride=1, padding=1, bias=False,
                            norm=FrozenBatchNorm2d(bottleneck_channels), act=ACT_RELU)
        self.conv3 = Conv2d(bottleneck_channels, out_channels, 1, bias=False, norm=FrozenBatchNorm2d(out_channels),
                            act=ACT_RELU)  # ReLU is applied after the residual add (fused)
        for layer in (self.conv1, self.conv2, self.conv3, self.shortcut):
            if layer is not None:
                c2_msra_fill(layer.weight)

    def forward(self, x):
        sc = self.shortcut(x) if self.shortcut is not None else x
        y = self.conv2(self.conv1(x))
        return self.conv3(y, res=sc)

    def forward_pair(self, x, a=None, nxt=None, frozen=False):
        """The block with its first layer possibly done already (`a` = conv1(x), produced by the previous block's launch) and its last
        layer possibly producing the NEXT block's first (`nxt`): returns (block output, nxt.conv1(block output) or None).
        conv3 + FrozenBN + residual + ReLU and the next conv1 + FrozenBN + ReLU as ONE launch (ops.conv2d_b2b, csrc/conv_xs_b2b.hip) where
        the pair has that form -- the block output is written once and not read back by the squeeze that follows it.
        frozen: the caller vouches that these layers do not train (the trainer's stem / res2 under FREEZE_AT 2), whatever the module mode."""
        from .layers import _CalibrationState

        sc = self.shortcut(x) if self.shortcut is not None else x
        b = self.conv2(self.conv1(x) if a is None else a)
        if nxt is not None and not _CalibrationState.active and (frozen or not self.training):
            pair = ops.conv2d_b2b(b, self.conv3.packed(), sc, nxt.conv1.packed())
            if pair is not None:
                return pair
        return self.conv3(b, res=sc), None


class ResNet(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        r = cfg.MODEL.RESNETS
        assert r.DEPTH == 50 and r.NORM == "FrozenBN" and r.STRIDE_IN_1X1 and r.NUM_GROUPS == 1 and r.RES5_DILATION == 1, \
            "only the ResNet-50 / FrozenBN / stride-in-1x1 backbone of the reference config is implemented"
        self.stem = BasicStem(3, r.STEM_OUT_CHANNELS)
        in_c, out_c, mid = r.STEM_OUT_CHANNELS, r.RES2_OUT_CHANNELS, r.NUM_GROUPS * r.WIDTH_PER_GROUP
        self._out_features = list(r.OUT_FEATURES)
        self._out_feature_channels, self._out_feature_strides = {}, {}
        stride_total = 4
        for name, nblk in (("res2", 3), ("res3", 4), ("res4", 6), ("res5", 3)):
            first_stride = 1 if name == "res2" else 2
            blocks = []
            for i in range(nblk):
                blocks.append(BottleneckBlock(in_c, out_c, mid, first_stride if i == 0 else 1))
                in_c = out_c
            for prev, blk in zip(blocks, blocks[1:]):  # (layer property: the pairs ops.conv2d_b2b can run as one launch)
                blk.conv1.b2b_second = (prev.conv3.in_channels, blk.conv1.out_channels) in ops.B2B_PAIRS
            self.add_module(name, nn.Sequential(*blocks))
            stride_total *= first_stride
            self._out_feature_channels[name], self._out_feature_strides[name] = out_c, stride_total
            out_c, mid = out_c * 2, mid * 2

    def forward_stage(self, name: str, x, frozen: bool = False):
        """One residual stage; block i's last layer may hand block i + 1 its first (BottleneckBlock.forward_pair)."""
        blocks = list(getattr(self, name))
        a = None
        for i, blk in enumerate(blocks):
            x, a = blk.forward_pair(x, a, blocks[i + 1] if i + 1 < len(blocks) else None, frozen=frozen)
        return x

    def forward(self, x4) -> Dict[str, torch.Tensor]:
        x = self.stem(x4)
        outs = {}
        for name in ("res2", "res3", "res4", "res5"):
            x = outs[name] = self.forward_stage(name, x)
        return outs

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n]) for n in self._out_features}


class FPN(nn.Module):
    """Top-down pyramid; lateral 1x1 convs add the nearest-x2-upsampled coarser level in their epilogue."""

    def __init__(self, cfg):
        super().__init__()
        f = cfg.MODEL.FPN
        assert f.NORM == "" and f.FUSE_TYPE == "sum", "reference config: FPN.NORM '' / FUSE_TYPE sum"
        self.bottom_up = ResNet(cfg)
        self.in_features = list(f.IN_FEATURES)
        shapes = self.bottom_up.output_shape()
        oc = f.OUT_CHANNELS
        self._levels = []
        for name in self.in_features:
            stage = int(name[3:])
            lat = Conv2d(shapes[name].channels, oc, 1, bias=True)
            out = Conv2d(oc, oc, 3, padding=1, bias=True)
            c2_xavier_fill(lat.weight, lat.bias)
            c2_xavier_fill(out.weight, out.bias)
            self.add_module(f"fpn_lateral{stage}", lat)
            self.add_module(f"fpn_output{stage}", out)
            self._levels.append(stage)
        self._out_features = [f"p{s}" for s in self._levels] + [f"p{self._levels[-1] + 1}"]
        self._out_feature_strides = {f"p{s}": 2 ** s for s in self._levels}
        self._out_feature_strides[self._out_features[-1]] = 2 ** (self._levels[-1] + 1)
        self._out_channels = oc
        self._size_divisibility = 2 ** self._levels[-1]
        self._zero_mean, self._unit_std = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)

    @property
    def size_divisibility(self):
        return self._size_divisibility

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_channels, stride=self._out_feature_strides[n]) for n in self._out_features}

    def forward_nhwc(self, x4: torch.Tensor) -> Dict[str, torch.Tensor]:
        """x4: [B,H,W,4] normalised frames -> {p2..p6} fp32 NHWC."""
        res = self.bottom_up(x4)
        top = self._levels[-1]
        prev = {top: getattr(self, f"fpn_lateral{top}")(res[f"res{top}"])}
        for s in reversed(self._levels[:-1]):  # the top-down chain (1x1 laterals + upsampled residual) is serial ...
            prev[s] = getattr(self, f"fpn_lateral{s}")(res[f"res{s}"], res=prev[s + 1], res_ups=True)
        # ... the 3x3 output convs are independent of each other: concurrent for 1-2 frame batches (streams.py), the stride-4
        # level on the current stream
        levels = list(self._levels)
        outs = run_branches([(lambda s=s: getattr(self, f"fpn_output{s}")(prev[s])) for s in levels],
                            concurrent=x4.is_cuda and x4.shape[0] <= SMALL_BATCH)
        feats = {f"p{s}": o for s, o in zip(levels, outs)}
        feats[f"p{top + 1}"] = ops.subsample2(feats[f"p{top}"])  # LastLevelMaxPool: max_pool2d(k=1, s=2)
        return feats

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Reference signature (planercnn.py:150): normalised NCHW batch -> dict of NCHW-shaped tensors."""
        x4 = getattr(x, "_a3d_nhwc4", None)  # set by PlaneRCNN.preprocess_image: already NHWC4
        if x4 is None:
            x4 = ops.preprocess_f32chw(x.contiguous().float(), self._zero_mean, self._unit_std)
        return {k: to_nchw_view(v) for k, v in self.forward_nhwc(x4).items()}


@BACKBONE_REGISTRY.register()
def build_resnet_fpn_backbone(cfg, input_shape=None):
    return FPN(cfg)


def build_backbone(cfg, input_shape=None):
    return BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)
