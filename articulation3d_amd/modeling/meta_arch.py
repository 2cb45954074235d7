"""`PlaneRCNN` meta-architecture (META_ARCH_REGISTRY).

Follows pkg/modeling/meta_arch/planercnn.py: constructor :26-58, `forward` :61-81 (eval),
`inference` :125-146, `inference_single` :148-184, `preprocess_image` :188-196, `_postprocess` :203-219.

Two entry points share the same kernels:
  * `forward(list[dict])` -- the reference's signature and output format (list of
    {"instances": Instances, "depth": Tensor});
  * `inference_batched(frames_u8)` -- the MI355X throughput path: a uint8 [B,H,W,3] BGR batch stays on the
    device end to end in fixed-size buffers (one small D2H read per batch), post-processing, mask paste and
    the plane-offset least squares are fused, and the result is the packed detection records that the
    frame-sharded all-gather ships to the host-side temporal optimiser.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional

import torch
from torch import nn

from .. import ops
from ..registry import META_ARCH_REGISTRY
from ..streams import SMALL_BATCH
from ..structures import Boxes, ImageList, Instances
from .backbone import build_backbone
from .depth_head import build_depth_head
from .layers import to_nchw_view
from .postprocessing import detector_postprocess  # noqa: F401  (reference-compatible export)
from .roi_heads.roi_heads import BatchedDetections, build_roi_heads
from .rpn import build_proposal_generator

__all__ = ["PlaneRCNN", "BatchedOutput", "build_model"]

POST_SCORE_THRESH = 0.1  # planercnn.py:217


@dataclass
class BatchedOutput:
    image_size: tuple
    det: BatchedDetections  # raw fixed-size detections + per-ROI head outputs (compact rows)
    depth: Optional[torch.Tensor]  # [B,H,W]
    keep: torch.Tensor  # [B,R] survives detector_postprocess
    boxes: torch.Tensor  # [B,R,4] clipped
    planes: torch.Tensor  # [B,R,3] normal*offset (PlaneRCNN_Branch.process output)
    area: torch.Tensor  # [B,R] pasted-mask pixel count
    masks: Optional[torch.Tensor]  # [B,R,H,W] uint8 or None
    records: torch.Tensor  # [B,R,798] packed detection records
    rec_count: torch.Tensor  # [B]
    proposals: Optional[tuple] = None


@META_ARCH_REGISTRY.register()
class PlaneRCNN(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.backbone = build_backbone(cfg)
        self.proposal_generator = build_proposal_generator(cfg, self.backbone.output_shape())
        self.roi_heads = build_roi_heads(cfg, self.backbone.output_shape())
        self.mask_threshold = cfg.MODEL.ROI_MASK_HEAD.MASK_THRESHOLD
        self.nms = cfg.MODEL.ROI_MASK_HEAD.NMS
        self.depth_head_on = cfg.MODEL.DEPTH_ON
        self.refine_on = cfg.MODEL.REFINE_ON
        self.axis_on = cfg.MODEL.AXIS_ON
        if self.depth_head_on:
            self.depth_head = build_depth_head(cfg)
        assert not self.refine_on, "REFINE_ON is false in every reference config (refine head out of scope)"
        self.vis_period = cfg.VIS_PERIOD
        self.input_format = cfg.INPUT.FORMAT
        assert len(cfg.MODEL.PIXEL_MEAN) == len(cfg.MODEL.PIXEL_STD) == 3
        self.pixel_mean = tuple(float(v) for v in cfg.MODEL.PIXEL_MEAN)
        self.pixel_std = tuple(float(v) for v in cfg.MODEL.PIXEL_STD)
        self._eval_gt_box = cfg.TEST.EVAL_GT_BOX
        self.to(self.device)
        self._freeze = cfg.MODEL.FREEZE
        for layers in self._freeze:
            final = self
            for l in layers.split("."):
                final = getattr(final, l)
            for params in final.parameters():
                params.requires_grad = False

    # ------------------------------------------------------------------ MI355X throughput path
    @torch.no_grad()
    def inference_batched(self, frames: torch.Tensor, want_masks: bool = False, given_boxes=None, source_rgb: bool = False,
                          resize_to=(480, 640)) -> BatchedOutput:
        """frames: uint8 [B,H,W,3] BGR (device) or float32 [B,3,H,W] 0-255 BGR.
        source_rgb=True: uint8 [B,Hs,Ws,3] straight from the reader -- RGB, any size -- and the reference loop's
        `cv2.resize(im, (640, 480))` + `im[:, :, ::-1]` (tools/inference.py:216-218) run fused with the normalisation in
        `a3d_preprocess_resize_u8`."""
        assert not self.training
        if frames.is_cuda and ops.DEFAULT_PRECISION == 3:
            # slots for the per-image maxima of this batch's activations (~200 feature maps + the per-ROI tensors), taken from a
            # chunk that is zero-filled HERE, on the main stream, before anything forks
            ops.amax_reserve(frames.shape[0] * (512 + 8 * int(self.proposal_generator.post_nms_topk[False])), frames.device)
        if source_rgb:
            assert frames.dtype == torch.uint8 and self.input_format == "BGR"
            B = frames.shape[0]
            H, W = int(resize_to[0]), int(resize_to[1])
            x4 = ops.preprocess_resize_u8(frames.contiguous(), self.pixel_mean, self.pixel_std, (H, W), swap_rb=True)
        elif frames.dtype == torch.uint8:
            B, H, W, _ = frames.shape
            x4 = ops.preprocess_u8hwc(frames.contiguous(), self.pixel_mean, self.pixel_std)
        else:
            B, _, H, W = frames.shape
            x4 = ops.preprocess_f32chw(frames.contiguous().float(), self.pixel_mean, self.pixel_std)
        assert H % self.backbone.size_divisibility == 0 and W % self.backbone.size_divisibility == 0, \
            "batched path expects frames already a multiple of 32 (the reference feeds 480x640)"
        hw = (H, W)
        feats = self.backbone.forward_nhwc(x4)
        # the RPN conv and the depth head's lateral conv read the same pyramid level: one Winograd input transform for both
        # (bit-identical results; -1.8 ms per 64 frames).  Not for 1-2 frame batches, whose RPN levels fork onto side streams.
        share = [f for f in feats.values() if f.is_cuda] if (self.depth_head_on and given_boxes is None and B > SMALL_BATCH) else []
        with ops.share_wino_input(share):
            return self._detect_on_features(feats, frames, B, hw, want_masks, given_boxes)

    # ------------------------------------------------------------------ the small-batch pass as a HIP graph (round 4)
    @torch.no_grad()
    def inference_graphed(self, frames: torch.Tensor, want_masks: bool = False) -> BatchedOutput:
        """`inference_batched(frames)` replayed from a captured HIP graph: for the per-frame loop of the reference
        (tools/inference.py:215-228 -> arti_vis.py:54-61), whose ~210 launches per frame the host issues one by one.  The first call per
        (shape, dtype, want_masks) warms up eagerly and captures ONE pass -- side-stream branches included -- into a graph with its own
        memory pool and its own zero-filled maxima slots; later calls copy the frame into the static input and replay.  Needs a pass
        without host reads: the ROI heads run in `fixed_rows` mode (roi_heads.py: head tensors sized for every detection slot).  The returned BatchedOutput's tensors live in the graph's pool: they are overwritten by the next call.  Same kernels, same
        bits as the eager pass (tests/test_gpu_bench.py).
        MEASURED AND NOT THE DEFAULT (round 4, tools/loop_bench.py): one frame 4.42 ms eager | 5.37 ms replayed, two frames 5.01 | 6.82 --
        the replay issues nothing from the host, but it pays the fixed-size head rows (5.43 ms eager in that mode) and gains nothing on
        top: the pass is bound by the latency of ~210 dependent small kernels on the GPU, not by their launches."""
        assert not self.training and frames.is_cuda
        rh = self.roi_heads
        B = frames.shape[0]
        R = rh.box_predictor.test_topk_per_image
        if B * R > rh.FIXED_ROWS_CAP:
            return self.inference_batched(frames, want_masks=want_masks)
        key = (tuple(frames.shape), frames.dtype, bool(want_masks), ops.DEFAULT_PRECISION, tuple(self.pinned_layers()))
        cache = self.__dict__.setdefault("_graphs", {})
        ent = cache.get(key)
        if ent is None:
            saved = rh.fixed_rows
            rh.fixed_rows = True
            try:
                static_in = frames.clone()
                for _ in range(2):  # warm-up: weight packs and their splits, LDS opt-ins, side streams, allocator
                    self.inference_batched(static_in, want_masks=want_masks)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                ops._amax_arena.pop(static_in.device, None)  # the pass gets maxima slots of its own, zero-filled INSIDE the graph
                with torch.cuda.graph(g):
                    out = self.inference_batched(static_in, want_masks=want_masks)
                ops._amax_arena.pop(static_in.device, None)  # (eager passes must not hand out slots of the graph's pool)
            finally:
                rh.fixed_rows = saved
            ent = cache[key] = (g, static_in, out)
        g, static_in, out = ent
        static_in.copy_(frames)
        g.replay()
        return out

    # ------------------------------------------------------------------ load-time precision audit (round 4)
    def _packables(self):
        """(qualified name, layer holder) of every conv / linear / deconv holder of the detection path, incl. the depth head's
        packers (plain attributes there, so that its state_dict keeps the reference's names only)."""
        from .layers import _Packable

        seen = {}
        for name, m in self.named_modules():
            if isinstance(m, _Packable):
                seen[id(m)] = (name, m)
        dh = getattr(self, "depth_head", None)
        if dh is not None:
            for kind in ("conv", "deconv"):
                for i, m in enumerate(getattr(dh, "_pk_" + kind, []), 1):
                    seen.setdefault(id(m), (f"depth_head.{kind}{i}", m))
        return list(seen.values())

    @torch.no_grad()
    def audit_precision(self, frames: torch.Tensor, pin: bool = True, source_rgb: bool = False, resize_to=(480, 640)):
        """Run `frames` (uint8, a few calibration frames of the deployment's kind) through the detector with every fp16x2 layer
        shadowed by its bf16x3 evaluation (ops.PrecisionAudit): each output element is held to the fp32-style one-term error law at its
        layer's own scale, and a layer that violates it anywhere is pinned to bf16x3 for the lifetime of the model -- per LAYER,
        statically, so results stay independent of batching.  Call once after loading a checkpoint (tools/inference.py does).
        Returns the audit (`.rows`: one record per layer launch; `.pinned()`).  Reference: the arithmetic this guards replaces the
        fp32 tensors of planercnn.py:125-184 on weights of the `exps/model_final.pth` kind (config.yaml:312)."""
        assert not self.training
        for name, m in self._packables():
            m._a3d_name = name
        audit = ops.PrecisionAudit(pin=pin)
        if ops.DEFAULT_PRECISION != 3:
            return audit  # (nothing to audit: bf16x3 and the fp32-input MFMA have no block exponents)
        with audit:
            self.inference_batched(frames, source_rgb=source_rgb, resize_to=resize_to)
            torch.cuda.synchronize()
        if pin:
            names = {r["layer"] for r in audit.pinned()}
            for name, m in self._packables():
                if name in names:
                    m.pin_precision = 2
        return audit

    def pinned_layers(self):
        return sorted(name for name, m in self._packables() if m.pin_precision == 2)

    def _detect_on_features(self, feats, frames, B, hw, want_masks, given_boxes) -> BatchedOutput:
        proposals = None
        if given_boxes is None:
            pb, pl, plv, ppos, pc = self.proposal_generator.forward_batched(feats, hw)
            proposals = (pb, pl, plv, ppos, pc)
            det = self.roi_heads.box_batched(feats, pb, pc, hw)
        else:  # forward_with_given_boxes entry (roi_heads.py:147): boxes [B,R,4], count [B]
            gb, gc = given_boxes
            det = BatchedDetections(gb, torch.ones(gb.shape[:2], device=gb.device), torch.zeros(gb.shape[:2], device=gb.device, dtype=torch.int32), gc, hw)
        # the live-ROI total starts its trip to the host here and is waited for after the depth decoder has been enqueued
        self.roi_heads.start_row_count(det)
        if self.depth_head_on and B <= self.small_batch_overlap and frames.is_cuda:
            # Small batches leave most of the 256 CUs idle (a 30x40 level is 10 tiles): the depth decoder runs on a second HIP
            # stream beside the ROI branch: +14% frames/s at 1-4 frames, +8% at 8, +4% at 16, +2.9% at 64 (fp16x2: a few hundred ROIs
            # are one partial round of the chip per head layer, and the decoder's tails fill it).
            main = torch.cuda.current_stream()
            if getattr(self, "_side_stream", None) is None:
                from ..streams import side

                self._side_stream = side(0)  # (the package's one pool of side streams: streams.py)
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(self._side_stream):
                self._side_stream.wait_event(ready)
                depth = self.depth_head.forward_nhwc(feats)
                done = torch.cuda.Event()
                done.record(self._side_stream)
            det = self.roi_heads.given_boxes_batched(feats, det)
            main.wait_event(done)
            depth.record_stream(main)
        else:
            depth = self.depth_head.forward_nhwc(feats) if self.depth_head_on else None
            det = self.roi_heads.given_boxes_batched(feats, det)
        return self._post_batched(det, depth, hw, want_masks, proposals)

    def _post_batched(self, det: BatchedDetections, depth, hw, want_masks, proposals) -> BatchedOutput:
        B, R = det.boxes.shape[:2]
        dev = det.boxes.device
        if det.total and det.mask_prob is not None:
            masks, planes, area, keep, boxes = ops.paste_lsq(
                det.boxes, det.scores, det.count, det.row_offset, det.mask_prob, det.pred_plane, depth, hw,
                post_score_thresh=POST_SCORE_THRESH, mask_thresh=self.mask_threshold, want_masks=want_masks)
            records, rec_count = ops.detections_pack(boxes, det.scores, det.classes, det.count, det.row_offset, keep, planes,
                                                     det.pred_rot_axis, det.pred_tran_axis, det.mask_prob, det.mask_prob.shape[-1])
        else:
            keep = torch.zeros((B, R), device=dev, dtype=torch.int32)
            boxes = torch.zeros((B, R, 4), device=dev)
            planes = torch.zeros((B, R, 3), device=dev)
            area = torch.zeros((B, R), device=dev, dtype=torch.int32)
            masks = torch.zeros((B, R, hw[0], hw[1]), device=dev, dtype=torch.uint8) if want_masks else None
            records = torch.zeros((B, R, ops.record_floats(28)), device=dev)
            rec_count = torch.zeros((B,), device=dev, dtype=torch.int32)
        return BatchedOutput(hw, det, depth, keep, boxes, planes, area, masks, records, rec_count, proposals)

    # ------------------------------------------------------------------ reference signatures
    def forward(self, batched_inputs):
        if not self.training:
            return self.inference(batched_inputs)
        return self.training_forward(batched_inputs)

    # ------------------------------------------------------------------ training branch (planercnn.py:83-123)
    def trainer(self, solver=None, precision: str = "bf16x3"):
        """The hand-written training step behind the reference's training-mode call: created on first use (it copies every
        trainable parameter into its flat buffer).  `articulation3d_amd.engine.build_optimizer` returns the optimiser bound to it."""
        if getattr(self, "_trainer", None) is None:
            from ..training import DetectorTrainer

            self._trainer = DetectorTrainer(self, solver, precision=precision)
        return self._trainer

    def training_forward(self, batched_inputs):
        """`model(batched_inputs)` in training mode, as detectron2's SimpleTrainer.run_step calls it (tools/train_net.py:84-104 ->
        planercnn.py:83-123): list of {"image": CHW uint8 BGR, "instances": Instances(gt_boxes, gt_classes)} -> the loss dict
        {loss_cls, loss_box_reg, loss_rpn_cls, loss_rpn_loc}.  There is no autograd graph: the call runs forward AND backward
        (DetectorTrainer.forward_backward) and leaves the gradients of the summed loss in the trainer's flat buffer; the returned
        scalars accept `.backward()` so the reference's loop body stays as it is, and `engine.build_optimizer(cfg, model).step()`
        applies the fused all-reduce + SGD launch.  Only the step1_bbox configuration (BASELINE configs[4]: box branch; mask /
        plane / axis / depth heads off) has a training path."""
        rh = self.roi_heads
        if self.depth_head_on or getattr(rh, "mask_on", False) or getattr(rh, "plane_on", False) or getattr(rh, "axis_on", False):
            raise NotImplementedError("training is implemented for config/step1_bbox.yaml (MASK_ON / PLANE_ON / AXIS_ON / DEPTH_ON false): "
                                      "the mask, plane, axis and depth losses of the later training stages are outside SURVEY.md 8f-1")
        assert "instances" in batched_inputs[0], "training needs ground-truth instances (planercnn.py:84-85)"
        imgs = [x["image"] for x in batched_inputs]
        assert all(tuple(t.shape) == tuple(imgs[0].shape) for t in imgs), "one image size per batch (the reference trains on 480x640 frames)"
        frames = torch.stack([t.to(self.device) for t in imgs]).permute(0, 2, 3, 1).contiguous()
        if frames.dtype != torch.uint8:
            frames = frames.round().clamp(0, 255).to(torch.uint8)  # (detectron2's mapper hands over uint8; tolerate float 0-255)
        gt_boxes = [x["instances"].gt_boxes.tensor.float() for x in batched_inputs]
        gt_classes = [x["instances"].gt_classes.long() for x in batched_inputs]
        tr = self.trainer()
        losses, _aux = tr.forward_backward(frames, gt_boxes, gt_classes, exchange=True)  # (optimizer.step() finishes the exchange)
        names = list(losses)
        outs = _StepLosses.apply(tr.autograd_anchor(), *[losses[k] for k in names])
        return dict(zip(names, outs))

    def train(self, mode: bool = True):
        """Leaving training mode writes the trainer's parameters back into the modules (inference packs its weights from them)."""
        if not mode and getattr(self, "_trainer", None) is not None and self.training:
            self.load_state_dict(self._trainer.export_state_dict(), strict=False)
        return super().train(mode)

    # batches up to this size run the depth decoder on a second HIP stream beside the ROI branch (0 disables: every kernel then runs alone)
    small_batch_overlap = int(os.environ.get("A3D_DEPTH_OVERLAP", "64"))
    fast_reference_path = True  # route uniform batches of the reference-signature call through inference_batched

    def _fast_path_ok(self, batched_inputs, do_postprocess) -> bool:
        if not (self.fast_reference_path and do_postprocess) or self._eval_gt_box or self.proposal_generator is None:
            return False
        rh = self.roi_heads
        if not (getattr(rh, "mask_on", False) and getattr(rh, "plane_on", False) and getattr(rh, "axis_on", False) and self.depth_head_on):
            return False
        shp = tuple(batched_inputs[0]["image"].shape)
        div = self.backbone.size_divisibility
        if len(shp) != 3 or shp[1] % div or shp[2] % div:
            return False
        for x in batched_inputs:
            if tuple(x["image"].shape) != shp or "instances" in x or "proposals" in x:
                return False
            if x.get("height", shp[1]) != shp[1] or x.get("width", shp[2]) != shp[2]:
                return False
        return True

    @torch.no_grad()
    def _inference_fast(self, batched_inputs):
        """Same outputs as inference_single + _postprocess (tests/test_gpu_parity.py::test_reference_api_matches_batched_path
        holds them bit-identical), from ONE pass of the fixed-size batched path: no per-stage Instances round trips and a
        single host read of the detection counts, instead of a synchronisation after every stage."""
        imgs = [x["image"].to(self.device, non_blocking=True) for x in batched_inputs]  # (stacking on the host costs ~9 ms per frame)
        frames = imgs[0][None] if len(imgs) == 1 else torch.stack(imgs)
        out = self.inference_batched(frames.float(), want_masks=True)  # CHW 0-255 BGR, any integer / float dtype
        hw = out.image_size
        cnt, keep = out.det.count.tolist(), out.keep.bool()
        ro = out.det.row_offset.tolist() if out.det.total else [0] * (len(cnt) + 1)
        res = []
        for b, n in enumerate(cnt):
            inst = Instances(hw)
            idx = keep[b, :n].nonzero().squeeze(1)
            inst.pred_boxes = Boxes(out.boxes[b, idx])
            inst.scores = out.det.scores[b, idx]
            inst.pred_classes = out.det.classes[b, idx].to(torch.int64)
            if out.det.total and out.masks is not None:
                inst.pred_masks = out.masks[b, idx].bool()
                inst.pred_plane = out.det.pred_plane[ro[b] + idx]
                inst.pred_rot_axis = out.det.pred_rot_axis[ro[b] + idx]
                inst.pred_tran_axis = out.det.pred_tran_axis[ro[b] + idx]
            else:
                dev = out.boxes.device
                inst.pred_masks = torch.zeros((0, hw[0], hw[1]), device=dev, dtype=torch.bool)
                inst.pred_plane, inst.pred_rot_axis = torch.zeros((0, 3), device=dev), torch.zeros((0, 3), device=dev)
                inst.pred_tran_axis = torch.zeros((0, 2), device=dev)
            res.append({"instances": inst, "depth": out.depth[b]})
        return res

    def inference(self, batched_inputs, detected_instances=None, do_postprocess=True):
        assert not self.training
        assert detected_instances is None
        if self._fast_path_ok(batched_inputs, do_postprocess):
            return self._inference_fast(batched_inputs)
        pred_instances, pred_depth = self.inference_single(batched_inputs, do_postprocess)
        for pre, d in zip(pred_instances, pred_depth):
            pre.update({"depth": d})
        return pred_instances

    @torch.no_grad()
    def inference_single(self, batched_inputs, do_postprocess=True):
        images = self.preprocess_image(batched_inputs)
        features = self.backbone(images.tensor)
        if self._eval_gt_box:
            gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
            for inst in gt_instances:
                inst.proposal_boxes = inst.gt_boxes
                inst.objectness_logits = torch.ones(len(inst.gt_boxes), device=self.device)
            proposals = gt_instances
        elif self.proposal_generator is not None:
            proposals, _ = self.proposal_generator(images, features, None)
        else:
            assert "proposals" in batched_inputs[0]
            proposals = [x["proposals"].to(self.device) for x in batched_inputs]
        pred_depth = [None] * len(proposals)
        if self.depth_head_on:
            pred_depth = self.depth_head(features, None)
        results, _ = self.roi_heads(images, features, proposals, None)
        return PlaneRCNN._postprocess(results, batched_inputs, images.image_sizes, mask_threshold=self.mask_threshold,
                                      nms=self.nms), pred_depth

    def preprocess_image(self, batched_inputs) -> ImageList:
        """planercnn.py:188-196: to(device), (x - mean)/std, pad to a multiple of 32, batch.
        The normalisation is the `a3d_preprocess_f32chw` kernel; `images.tensor` is an NCHW-shaped view of
        its NHWC4 output (the buffer itself rides along so the backbone does not convert back)."""
        imgs = [x["image"].to(self.device).float() for x in batched_inputs]
        sizes = [(int(t.shape[-2]), int(t.shape[-1])) for t in imgs]
        div = self.backbone.size_divisibility
        H = (max(s[0] for s in sizes) + div - 1) // div * div
        W = (max(s[1] for s in sizes) + div - 1) // div * div
        if all(s == (H, W) for s in sizes):
            x4 = ops.preprocess_f32chw(torch.stack(imgs).contiguous(), self.pixel_mean, self.pixel_std)
        else:  # ImageList.from_tensors pads the NORMALISED images with zeros
            x4 = torch.zeros((len(imgs), H, W, 4), device=self.device, dtype=torch.float32)
            for i, t in enumerate(imgs):
                x4[i, : t.shape[-2], : t.shape[-1]] = ops.preprocess_f32chw(t[None].contiguous(), self.pixel_mean, self.pixel_std)[0]
        tensor = to_nchw_view(x4)[:, :3]
        tensor._a3d_nhwc4 = x4
        return ImageList(tensor, sizes)

    @staticmethod
    def _postprocess(instances, batched_inputs, image_sizes, mask_threshold=0.5, nms=False):
        processed_results = []
        for results_per_image, input_per_image, image_size in zip(instances, batched_inputs, image_sizes):
            height = input_per_image.get("height", image_size[0])
            width = input_per_image.get("width", image_size[1])
            r = detector_postprocess(results_per_image, height, width, mask_threshold, box_score_threshold=POST_SCORE_THRESH, nms=nms)
            processed_results.append({"instances": r})
        return processed_results


class _StepLosses(torch.autograd.Function):
    """Gives the loss scalars of `training_forward` a `.backward()`: the gradients already sit in the trainer's flat buffer
    (computed for d(sum of the four losses)/d(parameters)), so backward only has to exist.  A caller that re-weights the losses
    before calling backward would silently not get what it asked for, hence the check that every upstream gradient is 1."""

    @staticmethod
    def forward(ctx, anchor, *losses):
        return tuple(l.detach().clone() for l in losses)

    @staticmethod
    def backward(ctx, *grads):
        for g in grads:
            if g is not None and float(g) != 1.0:
                raise RuntimeError("the hand-written backward computes the gradient of the plain SUM of the losses; re-weighted losses are not supported")
        return (None,) * (1 + len(grads))


def build_model(cfg):
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model
