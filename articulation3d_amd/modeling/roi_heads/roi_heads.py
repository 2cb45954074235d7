"""`PlaneRCNNROIHeads` (ROI_HEADS_REGISTRY): box -> mask -> plane -> axis on the kept boxes.

Follows pkg/modeling/roi_heads/roi_heads.py: constructor :26-83 (box / mask parts come from the
detectron2 StandardROIHeads parent), eval forward :118-130,
`forward_with_given_boxes` :147-165, `_forward_box` :167-207, `_forward_mask` :209-237,
`_forward_plane` :239-255, `_forward_axis` :257-273.

The plane and axis poolers are configured identically (14, ratio 0, ROIAlign) and receive the same boxes,
so their pooled tensors are identical: they are pooled once and shared (the reference pools twice).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
from torch import nn

from ... import ops
from ...streams import SMALL_BATCH, run_branches
from ...registry import ROI_HEADS_REGISTRY
from ...structures import Boxes, Instances, ShapeSpec
from ..box_head import FastRCNNOutputLayers, build_box_head
from ..layers import to_nhwc
from ..mask_head import build_mask_head
from ..poolers import ROIPooler
from .axis_head import build_axis_head
from .plane_head import build_plane_head

# heads run as concurrent branches on small batches (streams.SMALL_BATCH frames) and, beyond those, when the batch holds at most this
# many ROI rows in total (schedule only: same bits).  Default 0 since round 3: with the Winograd GEMMs' DMA ring a 64-frame clip at
# threshold 0.5 (276 rows) runs 44.36 ms per step with the heads one after the other and 44.43 ms with them side by side, and one after
# the other the plane and axis heads share one Winograd input transform.
HEADS_CONCURRENT_ROWS = int(os.environ.get("A3D_HEADS_CONCURRENT_ROWS", "0"))
# True: the box pooler writes its rows pre-split (a3d_roialign_desc.out_h2) and fc1 takes both operands by LDS-DMA ("conv_h2w_kernel xd").
# Same bits either way (tests/test_gpu_presplit.py).  Measured on the 64-frame clip (tools/fc1_bench.py, profiles/r04_fc1_dual_dma.txt):
# pooler 2.50 -> 3.31 ms (the row waits in LDS for its maximum: three 7-wave workgroups per CU, each with a serial split-and-store tail),
# fc1 4.05 -> 4.12-4.21 ms -- so the default stays the fp32 rows.  Why fc1 does not gain: DESIGN.md section 5 (round 4).
POOLER_PRESPLIT = os.environ.get("A3D_POOLER_PRESPLIT", "0") != "0"


class BatchedDetections:
    """Fixed-size device-side detections of a batch: slot (b, r), r < R; live slots r < count[b]."""

    def __init__(self, boxes, scores, classes, count, image_size):
        self.boxes, self.scores, self.classes, self.count = boxes, scores, classes, count
        self.image_size = image_size
        self.row_offset = None  # [B+1] compact row of (b, 0) in the per-ROI head outputs
        self.total = None  # host int: number of live detections in the batch
        self.mask_prob = self.pred_plane = self.pred_rot_axis = self.pred_tran_axis = None


@ROI_HEADS_REGISTRY.register()
class PlaneRCNNROIHeads(nn.Module):
    def __init__(self, cfg, input_shape: Dict[str, ShapeSpec]):
        super().__init__()
        r = cfg.MODEL.ROI_HEADS
        self.in_features = self.box_in_features = list(r.IN_FEATURES)
        self.num_classes = r.NUM_CLASSES
        self.mask_on = cfg.MODEL.MASK_ON
        scales = tuple(1.0 / input_shape[k].stride for k in self.in_features)
        in_channels = [input_shape[f].channels for f in self.in_features][0]
        # --- box branch (StandardROIHeads._init_box_head)
        b = cfg.MODEL.ROI_BOX_HEAD
        self.box_pooler = ROIPooler(b.POOLER_RESOLUTION, scales, b.POOLER_SAMPLING_RATIO, b.POOLER_TYPE)
        self.box_head = build_box_head(cfg, ShapeSpec(channels=in_channels, height=b.POOLER_RESOLUTION, width=b.POOLER_RESOLUTION))
        self.box_predictor = FastRCNNOutputLayers(cfg, self.box_head.output_shape)
        # --- mask branch (StandardROIHeads._init_mask_head)
        if self.mask_on:
            m = cfg.MODEL.ROI_MASK_HEAD
            self.mask_pooler = ROIPooler(m.POOLER_RESOLUTION, scales, m.POOLER_SAMPLING_RATIO, m.POOLER_TYPE)
            self.mask_head = build_mask_head(cfg, ShapeSpec(channels=in_channels, height=m.POOLER_RESOLUTION, width=m.POOLER_RESOLUTION))
        self._init_plane_head(cfg, input_shape, scales, in_channels)
        self._init_axis_head(cfg, input_shape, scales, in_channels)
        self._eval_gt_box = cfg.TEST.EVAL_GT_BOX
        self._refine_on = cfg.MODEL.REFINE_ON
        self._freeze = cfg.MODEL.FREEZE
        assert not self._refine_on, "REFINE_ON is false in every reference config (refine head out of scope)"

    def _init_plane_head(self, cfg, input_shape, scales, in_channels):  # roi_heads.py:38-59
        self.plane_on = cfg.MODEL.PLANE_ON
        if not self.plane_on:
            return
        p = cfg.MODEL.ROI_PLANE_HEAD
        self.plane_pooler = ROIPooler(p.POOLER_RESOLUTION, scales, p.POOLER_SAMPLING_RATIO, p.POOLER_TYPE)
        self.plane_head = build_plane_head(cfg, ShapeSpec(channels=in_channels, width=p.POOLER_RESOLUTION, height=p.POOLER_RESOLUTION))

    def _init_axis_head(self, cfg, input_shape, scales, in_channels):  # roi_heads.py:62-83
        self.axis_on = cfg.MODEL.AXIS_ON
        if not self.axis_on:
            return
        a = cfg.MODEL.ROI_AXIS_HEAD
        self.axis_pooler = ROIPooler(a.POOLER_RESOLUTION, scales, a.POOLER_SAMPLING_RATIO, a.POOLER_TYPE)
        self.axis_head = build_axis_head(cfg, ShapeSpec(channels=in_channels, width=a.POOLER_RESOLUTION, height=a.POOLER_RESOLUTION))

    # ------------------------------------------------------------------ batched (sync-light) path
    def box_batched(self, feats: Dict[str, torch.Tensor], prop_boxes, prop_count, img_hw) -> BatchedDetections:
        """_forward_box on fixed-size proposals [B,R,4] + count [B]."""
        lv = [feats[f] for f in self.box_in_features]
        # [B*R,7,7,C]; in the default arithmetic the pooler writes the rows pre-split for fc1 (both GEMM operands by LDS-DMA)
        pooled = self.box_pooler.forward_batched(lv, prop_boxes, prop_count, presplit=POOLER_PRESPLIT and len(self.box_head.fcs) > 0)
        pred = self.box_predictor(self.box_head(pooled))
        boxes, scores, classes, _pos, count = self.box_predictor.inference_batched(pred, prop_boxes, prop_count, img_hw)
        return BatchedDetections(boxes, scores, classes, count, img_hw)

    def start_row_count(self, det: BatchedDetections) -> BatchedDetections:
        """Row offsets of the compacted per-ROI tensors + an ASYNCHRONOUS copy of the live-ROI total into pinned host memory.
        The caller enqueues independent work (the depth decoder) before `given_boxes_batched` waits on the event, so the one
        host read of the batch does not drain the GPU queue."""
        R = det.boxes.shape[1]
        det.row_offset = ops.count_offsets(det.count, R)
        if self.fixed_rows and det.boxes.shape[0] * R <= self.FIXED_ROWS_CAP:
            return det  # (no host read in this mode: given_boxes_batched sizes the head tensors for every slot)
        if getattr(self, "_total_pin", None) is None:
            self._total_pin = torch.zeros(1, dtype=torch.int32, pin_memory=True)
        self._total_pin.copy_(det.row_offset[-1:], non_blocking=True)
        det._total_event = torch.cuda.Event()
        det._total_event.record()
        return det

    # Batches of at most this many detection SLOTS (frames x DETECTIONS_PER_IMAGE) can run their per-ROI heads without the step's one
    # host read (`fixed_rows`, off by default): the head tensors are sized for every slot, rows past the live total are zero-pooled and
    # never read (GEMM rows are independent; paste / pack go by the device-side counts); live rows keep their compact positions, so
    # the bits are those of the sized form.  It exists so that a pass can be captured as a HIP graph (PlaneRCNN.inference_graphed) --
    # and it is what that experiment costs: at one frame the sized heads see ~4 ROIs (a handful of plane-split workgroups per layer),
    # the fixed form 100, and the single-frame pass goes from 4.42 to 5.43 ms eager, 5.37 ms replayed from the graph
    # (tools/loop_bench.py, profiles/r04_loop_bench.txt): the host read is cheaper than the rows it saves.
    FIXED_ROWS_CAP = 256
    fixed_rows = False

    def given_boxes_batched(self, feats: Dict[str, torch.Tensor], det: BatchedDetections) -> BatchedDetections:
        """forward_with_given_boxes on fixed-size detections; one host read of the live-ROI total (none with `fixed_rows`)."""
        lv = [feats[f] for f in self.in_features]
        fixed = self.fixed_rows and det.boxes.shape[0] * det.boxes.shape[1] <= self.FIXED_ROWS_CAP
        if fixed:
            if det.row_offset is None:
                det.row_offset = ops.count_offsets(det.count, det.boxes.shape[1])
            det._total_event = None
            det.total = det.boxes.shape[0] * det.boxes.shape[1]
        else:
            if getattr(det, "_total_event", None) is None:
                self.start_row_count(det)
            det._total_event.synchronize()  # waits for the count only, not for the work enqueued after it
            det.total = int(self._total_pin[0])
            det._total_event = None
        rows = det.total
        if rows == 0:
            return det
        same_pool = self.plane_on and self.axis_on and self._same_pooler(self.plane_pooler, self.axis_pooler)
        pool = lambda p: p.forward_batched(lv, det.boxes, det.count, row_offset=det.row_offset, rows=rows, zero=fixed)
        shared = pool(self.plane_pooler) if self.plane_on else None
        # the three heads are independent: concurrent branches for 1-2 frame batches (streams.py), and for any batch whose
        # heads are single-round launches (few ROIs: each conv fills a fraction of the chip; measured +0.6% on the 64-frame clip)
        names, fns = [], []
        if self.mask_on:
            names.append("mask")
            fns.append(lambda: self.mask_head.forward_rows(pool(self.mask_pooler)))
        if self.plane_on:
            names.append("plane")
            fns.append(lambda: self.plane_head.forward_rows(shared))
        if self.axis_on:
            names.append("axis")
            fns.append(lambda: self.axis_head.forward_rows(shared if same_pool else pool(self.axis_pooler)))
        concurrent = det.boxes.is_cuda and (det.boxes.shape[0] <= SMALL_BATCH or rows <= HEADS_CONCURRENT_ROWS)
        # plane and axis heads read the same pooled tensor: their first 3x3 layers share one Winograd input transform
        with ops.share_wino_input([shared] if same_pool and not concurrent and shared.is_cuda else []):
            outs = dict(zip(names, run_branches(fns, concurrent=concurrent)))
        if "mask" in outs:
            det.mask_prob = outs["mask"]
        if "plane" in outs:
            det.pred_plane = outs["plane"]
        if "axis" in outs:
            det.pred_rot_axis, det.pred_tran_axis = outs["axis"]
        if self.plane_on and self.plane_head.keep_raw:  # checker hook: pre-normalisation head vectors (tests/test_gpu_e2e.py)
            det.raw_plane = self.plane_head.raw
        if self.axis_on and self.axis_head.keep_raw:
            det.raw_rot, det.raw_tran = self.axis_head.raw
        return det

    @staticmethod
    def _same_pooler(a: ROIPooler, b: ROIPooler) -> bool:
        return (a.output_size, a.sampling_ratio, a.aligned, a.scales) == (b.output_size, b.sampling_ratio, b.aligned, b.scales)

    # ------------------------------------------------------------------ reference signatures
    def forward(self, images, features, proposals, targets=None):
        """roi_heads.py:85-130 (eval branch): -> (list[Instances], {})."""
        del images
        if self.training:
            raise NotImplementedError("ROI-head losses (training) are outside the inference hot path (SURVEY.md 8f-1)")
        if self._eval_gt_box:
            pred_instances = [Instances(p.image_size) for p in proposals]
            for ins, p in zip(pred_instances, proposals):
                ins.pred_boxes = p.gt_boxes
                ins.scores = torch.ones(len(p.gt_boxes), device=p.gt_boxes.device)  # the reference hard-codes "cuda" (:123)
                ins.pred_classes = p.gt_classes
        else:
            pred_instances = self._forward_box(features, proposals)
        pred_instances = self.forward_with_given_boxes(features, pred_instances)
        return pred_instances, {}

    def _forward_box(self, features, proposals: List[Instances]) -> List[Instances]:
        feats = {f: to_nhwc(features[f]) for f in self.box_in_features}
        dev = feats[self.box_in_features[0]].device
        hw = proposals[0].image_size
        B = len(proposals)
        R = max(1, max(len(p) for p in proposals))
        boxes = torch.zeros((B, R, 4), device=dev, dtype=torch.float32)
        for i, p in enumerate(proposals):
            if len(p):
                boxes[i, : len(p)] = p.proposal_boxes.tensor
        count = torch.tensor([len(p) for p in proposals], device=dev, dtype=torch.int32)
        det = self.box_batched(feats, boxes, count, hw)
        out = []
        for b, n in enumerate(det.count.tolist()):
            inst = Instances(hw)
            inst.pred_boxes = Boxes(det.boxes[b, :n])
            inst.scores = det.scores[b, :n]
            inst.pred_classes = det.classes[b, :n].to(torch.int64)
            out.append(inst)
        return out

    def forward_with_given_boxes(self, features, instances: List[Instances]) -> List[Instances]:
        """roi_heads.py:147-165."""
        assert not self.training
        assert instances[0].has("pred_boxes") and instances[0].has("pred_classes")
        feats = {f: to_nhwc(features[f]) for f in self.in_features}
        dev = feats[self.in_features[0]].device
        hw = instances[0].image_size
        B = len(instances)
        R = max(1, max(len(i) for i in instances))
        boxes = torch.zeros((B, R, 4), device=dev, dtype=torch.float32)
        for i, ins in enumerate(instances):
            if len(ins):
                boxes[i, : len(ins)] = ins.pred_boxes.tensor
        count = torch.tensor([len(i) for i in instances], device=dev, dtype=torch.int32)
        det = BatchedDetections(boxes, None, None, count, hw)
        det = self.given_boxes_batched(feats, det)
        n = [len(i) for i in instances]
        zeros = lambda *s: torch.zeros(s, device=dev, dtype=torch.float32)
        if self.mask_on:
            mp = det.mask_prob if det.mask_prob is not None else zeros(0, 28, 28)
            for ins, m in zip(instances, mp.split(n, 0)):
                ins.pred_masks = m[:, None]  # (D,1,28,28) as mask_rcnn_inference
        if self.plane_on:
            pl = det.pred_plane if det.pred_plane is not None else zeros(0, 3)
            for ins, p in zip(instances, pl.split(n, 0)):
                ins.pred_plane = p
        if self.axis_on:
            ra = det.pred_rot_axis if det.pred_rot_axis is not None else zeros(0, 3)
            ta = det.pred_tran_axis if det.pred_tran_axis is not None else zeros(0, 2)
            for ins, a, t in zip(instances, ra.split(n, 0), ta.split(n, 0)):
                ins.pred_rot_axis = a
                ins.pred_tran_axis = t
        return instances


def build_roi_heads(cfg, input_shape):
    return ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, input_shape)
