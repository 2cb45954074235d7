"""`ROIPooler`: FPN level assignment + ROIAlign in one kernel launch.

Same constructor as the detectron2 class the reference instantiates at
pkg/modeling/roi_heads/roi_heads.py:50-55,74-79 (output_size, scales, sampling_ratio, pooler_type)."""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch
from torch import nn

from .. import ops
from ..structures import Boxes
from .layers import to_nhwc


class ROIPooler(nn.Module):
    def __init__(self, output_size, scales: Sequence[float], sampling_ratio: int, pooler_type: str,
                 canonical_box_size: int = 224, canonical_level: int = 4):
        super().__init__()
        if isinstance(output_size, (tuple, list)):
            assert output_size[0] == output_size[1]
            output_size = output_size[0]
        assert pooler_type in ("ROIAlign", "ROIAlignV2"), f"unsupported pooler type {pooler_type}"
        assert canonical_box_size == 224 and canonical_level == 4, "the kernel hard-codes the FPN paper's k0=4, s0=224"
        self.output_size = int(output_size)
        self.scales = [float(s) for s in scales]
        self.sampling_ratio = int(sampling_ratio)
        self.aligned = pooler_type == "ROIAlignV2"
        min_level = -math.log2(self.scales[0])
        assert math.isclose(min_level, int(min_level)) and int(min_level) == 2, "pyramid must start at stride 4 (p2)"

    def forward_batched(self, feats_nhwc: List[torch.Tensor], boxes: torch.Tensor, count: Optional[torch.Tensor], *,
                        row_offset: Optional[torch.Tensor] = None, rows: Optional[int] = None, presplit: bool = False, zero: bool = False):
        """boxes [B,R,4] fixed-size slots, count [B] live slots -> [rows, P, P, C] NHWC bins (presplit: the rows as the fp16 planes
        of the default arithmetic, for a Linear consumer -- ops.roi_align_fpn)."""
        return ops.roi_align_fpn(feats_nhwc, self.scales, boxes, count, self.output_size, self.sampling_ratio,
                                 self.aligned, row_offset=row_offset, rows=rows, presplit=presplit, zero=zero)

    def forward(self, x: List[torch.Tensor], box_lists: List[Boxes]):
        """Reference signature: list of NCHW level features + per-image Boxes -> [sum(K_i), C, P, P]."""
        feats = [to_nhwc(t) for t in x]
        dev = feats[0].device
        B = len(box_lists)
        R = max(1, max(len(b) for b in box_lists))
        boxes = torch.zeros((B, R, 4), device=dev, dtype=torch.float32)
        for i, bl in enumerate(box_lists):
            if len(bl):
                boxes[i, : len(bl)] = bl.tensor.to(dev)
        count = torch.tensor([len(b) for b in box_lists], device=dev, dtype=torch.int32)
        off = ops.count_offsets(count, R)
        total = int(sum(len(b) for b in box_lists))
        out = self.forward_batched(feats, boxes, count, row_offset=off, rows=max(total, 1))
        return out[:total].permute(0, 3, 1, 2)
