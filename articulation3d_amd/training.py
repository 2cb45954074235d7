"""The detector TRAINING step on MI355X (BASELINE configs[4]: config/step1_bbox.yaml).

Replaces what runs when the reference trains (tools/train_net.py:84-117 -> detectron2 DefaultTrainer):
  * `PlaneRCNN.forward` with `self.training` (pkg/modeling/meta_arch/planercnn.py:83-123): backbone, RPN with losses,
    `PlaneRCNNROIHeads.forward` training branch (pkg/modeling/roi_heads/roi_heads.py:93-117) whose only loss source under
    step1_bbox.yaml (MASK / PLANE / AXIS / DEPTH off) is `_forward_box` (:190-204);
  * autograd's backward pass;  * torch.optim.SGD + WarmupMultiStepLR;  * DistributedDataParallel's gradient all-reduce.

Design (MI355X-first, no autograd engine):
  * All trainable parameters (41.08 M fp32: res3-res5, FPN, RPN head, box head, predictor; stem + res2 are frozen by
    FREEZE_AT 2 and FrozenBN never trains) live in ONE flat device buffer in the packed layout the conv kernel consumes
    ([Cout][KH][KW][Cin]); gradients and momentum are two more flat buffers of the same shape.  Data-parallel training
    is therefore ONE RCCL all-reduce of 164 MB per step and ONE fused SGD launch.
  * Forward = the inference kernels (same launches, same results) with activations kept.
  * Backward is written out explicitly: data gradients run through the SAME fp32-MFMA conv kernel with transposed /
    flipped filters (a3d_weight_transpose) and the ReLU mask + shortcut add fused in its epilogue (`gate`, `res`);
    weight gradients through the pixel-reduction GEMM (a3d_conv_wgrad_nhwc_f32); ROIAlign backward by atomics.
  * Labelling (Matcher) AND the random sub-sampling (256 anchors, 512 ROIs per image) run on the GPU (counter-based
    RNG: a3d_sample_labels / a3d_sample_rois), all per-image counts stay in device vectors and tensors have fixed shapes
    (dead slots are zero rows without loss or gradient): the step never waits for the host.  Parity tests read the drawn
    index sets back and hand them to the oracle.
Precision: fp32-grade ("bf16x3") by default; `precision="fp32"` is the fp32-input MFMA; `precision="bf16"` is the reference config's autocast arithmetic (bf16 MFMA, fp32 accumulation,
fp32 master weights) on every trainable layer, with the ResNet stages' activations and gradients STORED as bf16 (round 3:
a3d_conv_desc.io_bf16 / a3d_wgrad_desc.io_bf16; `storage="fp32"` keeps them fp32) -- since round 4 also the FPN lateral sums, the RPN
hidden maps, the box head's hidden rows and the gradients flowing back through them -- and a bf16 gradient all-reduce payload.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib, ops, train_ops as T
from .parallel import GradientExchange, allreduce_gradients
from .ops import ACT_NONE, ACT_RELU, PackedConv

FPN_STRIDES = {"p2": 4, "p3": 8, "p4": 16, "p5": 32, "p6": 64}
# One launch for all data-gradient filters and one for all weight-gradient slice reductions of a step instead of one per layer (round 4;
# "0": the per-layer launches -- the same bits either way, tests/test_gpu_training.py).
BATCHED_LAUNCHES = os.environ.get("A3D_TRAIN_BATCHED", "1") != "0"
WGRAD_SIDE_STREAM = os.environ.get("A3D_TRAIN_WGRAD_STREAM", "1") != "0"  # weight gradients on a side stream beside the data-gradient chain (same bits)
RPN_BWD_STREAM = os.environ.get("A3D_TRAIN_RPN_STREAM", "1") != "0"  # the RPN head's backward as soon as its loss exists, on a second stream (same bits)
# The gradient exchange of `step()`: "1" (default) = in segments under the backward pass (parallel.GradientExchange), "0" = one collective
# behind it (parallel.allreduce_gradients; the same bits at world 2), "force" = the segmented form even at world 1 (measurement only),
# "late" / "force-late" = the same segments, every one announced only BEHIND the whole backward pass (the equality partner of the tests: a
# segment announced too early shows up as a bit that differs from this form's).
GRAD_OVERLAP = os.environ.get("A3D_TRAIN_GRAD_OVERLAP", "1")
RES_STAGES = (("res3", 4, 128, 512), ("res4", 6, 256, 1024), ("res5", 3, 512, 2048))  # name, blocks, mid, out
GT_LOGIT = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))


@dataclass
class SolverCfg:
    """detectron2 defaults selected by config/step1_bbox.yaml (its _BASE_ is commented out)."""
    rpn_batch_per_image: int = 256
    rpn_positive_fraction: float = 0.5
    rpn_iou_thresholds: Tuple[float, float] = (0.3, 0.7)
    rpn_pre_topk_train: int = 2000  # step1_bbox.yaml:21 (2048-slot selection groups, NMS words in global memory)
    rpn_post_topk_train: int = 1000
    rpn_nms_thresh: float = 0.7
    roi_batch_per_image: int = 512
    roi_positive_fraction: float = 0.25
    roi_iou_threshold: float = 0.5
    num_classes: int = 2
    rpn_weights: Tuple[float, ...] = (1.0, 1.0, 1.0, 1.0)
    box_weights: Tuple[float, ...] = (10.0, 10.0, 5.0, 5.0)
    base_lr: float = 0.001
    momentum: float = 0.9
    weight_decay: float = 1e-4
    warmup_iters: int = 1000
    warmup_factor: float = 0.001
    steps: Tuple[int, ...] = (210000, 250000)
    gamma: float = 0.1
    max_gt: int = 16


def lr_at(it: int, s: SolverCfg) -> float:
    """WarmupMultiStepLR with linear warm-up."""
    f = 1.0
    if it < s.warmup_iters:
        a = it / s.warmup_iters
        f = s.warmup_factor * (1 - a) + a
    return s.base_lr * f * s.gamma ** sum(1 for m in s.steps if m <= it)


def subsample_labels(labels: torch.Tensor, num: int, pos_frac: float, bg_label: int, gen: torch.Generator):
    """detectron2.modeling.sampling.subsample_labels on a CPU label vector with an explicit generator."""
    positive = ((labels != -1) & (labels != bg_label)).nonzero().squeeze(1)
    negative = (labels == bg_label).nonzero().squeeze(1)
    num_pos = min(positive.numel(), int(num * pos_frac))
    num_neg = min(negative.numel(), num - num_pos)
    p1 = torch.randperm(positive.numel(), generator=gen)[:num_pos]
    p2 = torch.randperm(negative.numel(), generator=gen)[:num_neg]
    return positive[p1], negative[p2]


@dataclass
class _Layer:
    """One trainable conv / linear: views into the flat parameter, gradient and scratch buffers."""
    name: str
    rows: int  # GEMM columns of the forward layer (output channels, padded)
    cin: int
    k: int
    stride: int
    pad: int
    act: int
    w: torch.Tensor = None   # [rows, k*k*cin]
    dw: torch.Tensor = None
    b: Optional[torch.Tensor] = None  # [rows] or None
    db: Optional[torch.Tensor] = None
    wt: torch.Tensor = None  # transposed filter scratch [cin, k*k*rows]
    wb: Optional[torch.Tensor] = None   # bf16 step: w rounded to bf16 (view of the flat bf16 copy of the parameters, refreshed once per step)
    wtb: Optional[torch.Tensor] = None  # ... and the data-gradient filter
    U: Optional[torch.Tensor] = None   # Winograd-domain forward filter (3x3 only)
    Ut: Optional[torch.Tensor] = None  # Winograd-domain data-gradient filter
    U3: Optional[torch.Tensor] = None   # precision "bf16x3": U / Ut split into three bf16 planes, refreshed once per step by
    Ut3: Optional[torch.Tensor] = None  # _prepare_filters (a fresh PackedConv per launch would otherwise re-split on every call)
    scale: Optional[torch.Tensor] = None  # folded FrozenBN (constants)
    shift: Optional[torch.Tensor] = None
    sources: List[Tuple[str, int, int]] = field(default_factory=list)  # (state_dict prefix, first row, rows) fused heads

    def fwd(self) -> PackedConv:
        K = self.k * self.k * self.cin
        p = PackedConv(self.w, self.scale, self.shift if self.b is None else self.b, self.k, self.k, self.stride, self.pad, self.cin,
                       self.rows, K, self.act)
        if self.U is not None:
            p.w_wino = self.U
            p.w_wino_x3 = self.U3
        p.w_b16 = self.wb
        return p

    def bwd(self) -> PackedConv:
        """The data-gradient conv: input channels = forward rows, stride 1 (stride-2 layers scatter afterwards)."""
        K = self.k * self.k * self.rows
        p = PackedConv(self.wt, None, None, self.k, self.k, 1, self.k - 1 - self.pad, self.rows, self.cin, K, ACT_NONE)
        p.w_b16 = self.wtb
        if self.Ut is not None:
            p.w_wino = self.Ut
            p.w_wino_x3 = self.Ut3
        return p


class DetectorTrainer:
    """One training step of the step1_bbox configuration: losses, gradients and the SGD update, all on the device.

    `model` is the product PlaneRCNN (its frozen stem / res2 modules are used as they are; everything trainable is
    copied into the flat buffer at construction and written back by `export_state_dict`)."""

    def __init__(self, model, solver: Optional[SolverCfg] = None, seed: int = 2020, process_group=None, precision: str = "bf16x3",
                 grad_payload: Optional[str] = None, storage: Optional[str] = None, grad_overlap: Optional[str] = None):
        """precision: "bf16x3" (the default since round 3: fp32-GRADE products from exact 3-way bf16 operand splits, six bf16 MFMAs per
        fp32 multiply-add -- no block exponents, so filters that change every step need no maxima; 140 | 319 images/s at 2 | 16
        images per GPU against 123 | 270 of the fp32-input MFMA), "fp32" (fp32-input MFMA everywhere) or "bf16" -- the reference's
        autocast setting: every trainable conv / linear multiplies bf16-rounded operands on the bf16 MFMA with fp32 accumulation
        (forward, data and weight gradients); master weights, losses and the optimiser stay fp32."""
        assert precision in ("fp32", "bf16", "bf16x3")
        # "bf16x3": fp32-grade arithmetic on the bf16 pipe for every forward / data-gradient launch -- csrc/conv_bf16x3.hip for the
        # direct layers, the split-operand Winograd GEMM (csrc/conv_wino.hip 2x) for the 3x3 ones, whose filters are split once per
        # step in _prepare_filters -- and for every weight gradient
        # gradient all-reduce payload: bf16 in the bf16 step (configs[4]: DDP's bf16_compress_hook semantics), fp32 otherwise
        self.grad_payload = grad_payload or ("bf16" if precision == "bf16" else "fp32")
        self.grad_overlap = GRAD_OVERLAP if grad_overlap is None else str(grad_overlap)  # "1" | "0" | "force" (see GRAD_OVERLAP)
        self._xchg: Optional[GradientExchange] = None
        self._xchg_live = False
        # storage of the ResNet stages' activations and gradients (res3-res5: the bulk of the step's activation bytes): bf16 in the bf16
        # step -- what autocast itself keeps (conv outputs are bf16 tensors under torch.autocast) -- unless storage="fp32" is asked for.
        # FPN / RPN / box-head tensors, the pyramid gradients (float atomics) and everything a loss kernel reads stay fp32.
        self.storage = storage or ("bf16" if precision == "bf16" else "fp32")
        assert self.storage in ("fp32", "bf16") and (self.storage == "fp32" or precision == "bf16"), "bf16 storage belongs to the bf16 step"
        self._st = torch.bfloat16 if self.storage == "bf16" else None
        self.prec = {"fp32": 0, "bf16": 1, "bf16x3": "bf16x3"}[precision]
        self.wgrad_prec = {"fp32": 0, "bf16": 1, "bf16x3": 2}[precision]
        self.s = solver or SolverCfg()
        self.model = model
        self.dev = next(model.parameters()).device
        self.seed = int(seed)
        self.pg = process_group
        self.iter = 0
        sd = {k: v.detach().float() for k, v in model.state_dict().items()}
        self.pixel_mean, self.pixel_std = model.pixel_mean, model.pixel_std
        self.cell_anchors = model.proposal_generator.anchor_generator.cell_anchors
        self.strides = [FPN_STRIDES[n] for n in ("p2", "p3", "p4", "p5", "p6")]
        self._build_layers(sd)
        self._anchors_cache: Dict[Tuple[int, int], torch.Tensor] = {}

    # ------------------------------------------------------------------------------------------ parameters
    def _build_layers(self, sd):
        L: Dict[str, _Layer] = {}
        bu = "backbone.bottom_up."

        def bn_fold(prefix):
            n = prefix + ".norm."
            scale, shift = ops.fold_bn(sd[n + "weight"], sd[n + "bias"], sd[n + "running_mean"], sd[n + "running_var"], 1e-5)
            return scale.to(self.dev).contiguous(), shift.to(self.dev).contiguous()

        cin = 256
        for name, nblk, mid, cout in RES_STAGES:
            for i in range(nblk):
                p = f"{bu}{name}.{i}."
                s = 2 if i == 0 else 1
                specs = [("conv1", mid, cin, 1, s, 0, ACT_RELU), ("conv2", mid, mid, 3, 1, 1, ACT_RELU), ("conv3", cout, mid, 1, 1, 0, ACT_RELU)]
                if i == 0:
                    specs.append(("shortcut", cout, cin, 1, s, 0, ACT_NONE))
                for cn, rows, ci, k, st, pad, act in specs:
                    ly = _Layer(p + cn, rows, ci, k, st, pad, act)
                    ly.scale, ly.shift = bn_fold(p + cn)
                    L[ly.name] = ly
                cin = cout
        for l, c in ((2, 256), (3, 512), (4, 1024), (5, 2048)):
            L[f"backbone.fpn_lateral{l}"] = _Layer(f"backbone.fpn_lateral{l}", 256, c, 1, 1, 0, ACT_NONE)
            L[f"backbone.fpn_output{l}"] = _Layer(f"backbone.fpn_output{l}", 256, 256, 3, 1, 1, ACT_NONE)
        rp = "proposal_generator.rpn_head."
        L[rp + "conv"] = _Layer(rp + "conv", 256, 256, 3, 1, 1, ACT_RELU)
        L[rp + "pred"] = _Layer(rp + "pred", 32, 256, 1, 1, 0, ACT_NONE,
                                sources=[(rp + "objectness_logits", 0, 3), (rp + "anchor_deltas", 3, 12)])
        bh = "roi_heads.box_head."
        L[bh + "fc1"] = _Layer(bh + "fc1", 1024, 256 * 49, 1, 1, 0, ACT_RELU)
        L[bh + "fc2"] = _Layer(bh + "fc2", 1024, 1024, 1, 1, 0, ACT_RELU)
        bp = "roi_heads.box_predictor."
        K = self.s.num_classes
        L[bp + "pred"] = _Layer(bp + "pred", 32, 1024, 1, 1, 0, ACT_NONE, sources=[(bp + "cls_score", 0, K + 1), (bp + "bbox_pred", K + 1, 4 * K)])
        self.layers = L
        # (side streams 0 and 1 of the package's one pool -- streams.side: which hardware queue a stream lands on follows from the order
        # in which a process first uses its streams, and the step's time with it)
        from .streams import side
        self._wg_stream = side(0, self.dev) if WGRAD_SIDE_STREAM else None
        self._rpn_stream = side(1, self.dev) if RPN_BWD_STREAM else None
        has_bias = lambda ly: ly.scale is None
        n = sum(ly.rows * ly.k * ly.k * ly.cin + (ly.rows if has_bias(ly) else 0) for ly in L.values())
        n = (n + 3) // 4 * 4
        self.params = torch.zeros(n, device=self.dev)
        self.grads = torch.zeros(n, device=self.dev)
        self.momentum = torch.zeros(n, device=self.dev)
        nw = sum(ly.rows * ly.k * ly.k * ly.cin for ly in L.values())
        self._wt = torch.empty(nw, device=self.dev)
        nu = sum(16 * ly.rows * ly.cin for ly in L.values() if ly.k == 3)
        self._U = torch.empty(nu, device=self.dev)
        self._Ut = torch.empty(nu, device=self.dev)
        # bf16 step: the parameters and the data-gradient filters ALSO as bf16 (rounded once per step in _prepare_filters: two launches), in
        # the same flat layouts -- what csrc/conv_bf16w.hip DMAs (every filter starts at a multiple of 8 elements: 16-byte aligned)
        self._p16 = torch.empty(n, device=self.dev, dtype=torch.bfloat16) if self.prec == 1 else None
        self._wt16 = torch.empty(nw, device=self.dev, dtype=torch.bfloat16) if self.prec == 1 else None
        off = woff = uoff = 0
        for ly in L.values():
            nwl = ly.rows * ly.k * ly.k * ly.cin
            ly.w = self.params[off:off + nwl].view(ly.rows, -1)
            ly.dw = self.grads[off:off + nwl].view(ly.rows, -1)
            if self._p16 is not None:
                assert off % 8 == 0 and woff % 8 == 0
                ly.wb = self._p16[off:off + nwl].view(ly.rows, -1)
                ly.wtb = self._wt16[woff:woff + nwl].view(ly.cin, -1)
            off += nwl
            if has_bias(ly):
                ly.b, ly.db = self.params[off:off + ly.rows], self.grads[off:off + ly.rows]
                off += ly.rows
            ly.wt = self._wt[woff:woff + nwl].view(ly.cin, -1)
            woff += nwl
            if ly.k == 3:
                nul = 16 * ly.rows * ly.cin
                ly.U = self._U[uoff:uoff + nul].view(16, ly.rows, ly.cin)
                ly.Ut = self._Ut[uoff:uoff + nul].view(16, ly.cin, ly.rows)
                uoff += nul
        # Gradient-exchange segments in the order the backward pass completes them.  The flat buffer is in forward order (res3, res4, res5,
        # FPN, RPN head, box head), the backward pass walks it back to front, so the segments are contiguous ranges taken from the end.
        first = {}
        o = 0
        for ly in L.values():
            first.setdefault("box" if ly.name.startswith("roi_heads.") else "fpn" if not ly.name.startswith("backbone.bottom_up.")
                             else ly.name.split(".")[2], o)
            o += ly.rows * ly.k * ly.k * ly.cin + (ly.rows if has_bias(ly) else 0)
        cuts = [first["res3"], first["res4"], first["res5"], first["box"], n]
        assert cuts[0] == 0 and first["res5"] < first["fpn"] < first["box"] and all(a < b and a % 4 == 0 for a, b in zip(cuts, cuts[1:])), cuts
        # box head (13.9 M, final 0.3 ms into the backward pass) | res5 + FPN + RPN head (19.1 M) | res4 (7.1 M) | res3 (1.2 M: the only
        # piece whose transfer nothing can hide -- 2.4 MB of bf16).  Four, not more: at the reference's 2 images per GPU every collective
        # costs ~70 us of host time on a step whose host side is nearly as long as its GPU side.
        self.grad_segments = [(cuts[i], cuts[i + 1]) for i in (3, 2, 1, 0)]
        self.load_state_dict(sd)

    def _views(self, ly: _Layer, sd_like: bool, buf_w, buf_b):
        """(state_dict name, tensor in torch layout) pairs of one layer, read from packed buffers."""
        out = []
        if ly.sources:
            for prefix, r0, nr in ly.sources:
                w = buf_w[r0:r0 + nr]
                out.append((prefix + ".weight", w.reshape(nr, ly.cin, 1, 1) if "roi_heads" not in prefix else w.reshape(nr, ly.cin)))
                out.append((prefix + ".bias", buf_b[r0:r0 + nr]))
            return out
        if ly.name.endswith("fc1"):
            w = buf_w.view(ly.rows, 7, 7, 256).permute(0, 3, 1, 2).reshape(ly.rows, -1)
        elif ly.name.endswith("fc2"):
            w = buf_w
        else:
            w = buf_w.view(ly.rows, ly.k, ly.k, ly.cin).permute(0, 3, 1, 2)
        out.append((ly.name + ".weight", w))
        if buf_b is not None:
            out.append((ly.name + ".bias", buf_b))
        return out

    @torch.no_grad()
    def load_state_dict(self, sd):
        for ly in self.layers.values():
            if ly.sources:
                ly.w.zero_()
                ly.b.zero_()
                for prefix, r0, nr in ly.sources:
                    ly.w[r0:r0 + nr] = sd[prefix + ".weight"].reshape(nr, -1).to(self.dev)
                    ly.b[r0:r0 + nr] = sd[prefix + ".bias"].to(self.dev)
                continue
            w = sd[ly.name + ".weight"].to(self.dev)
            if ly.name.endswith("fc1"):
                w = w.view(ly.rows, 256, 7, 7).permute(0, 2, 3, 1)
            elif w.dim() == 4:
                w = w.permute(0, 2, 3, 1)
            ly.w.copy_(w.reshape(ly.rows, -1))
            if ly.b is not None:
                ly.b.copy_(sd[ly.name + ".bias"].to(self.dev))

    def export_state_dict(self) -> Dict[str, torch.Tensor]:
        """Trainable parameters under their detectron2 names / layouts (checkpoint format)."""
        return {k: v.detach().clone().contiguous() for ly in self.layers.values() for k, v in self._views(ly, True, ly.w, ly.b)}

    def export_grads(self) -> Dict[str, torch.Tensor]:
        return {k: v.detach().clone().contiguous() for ly in self.layers.values() for k, v in self._views(ly, True, ly.dw, ly.db)}

    # ------------------------------------------------------------------------------------------ helpers
    def _anchors(self, feat_hw) -> torch.Tensor:
        key = tuple(feat_hw)
        if key not in self._anchors_cache:
            per = []
            for (h, w), s, cell in zip(feat_hw, self.strides, self.cell_anchors):
                sx = torch.arange(0, w * s, s, dtype=torch.float32)
                sy = torch.arange(0, h * s, s, dtype=torch.float32)
                yy, xx = torch.meshgrid(sy, sx, indexing="ij")
                shifts = torch.stack((xx, yy, xx, yy), -1).reshape(-1, 1, 4)
                per.append((shifts + cell.view(1, -1, 4)).reshape(-1, 4))
            self._anchors_cache[key] = torch.cat(per, 0).contiguous().to(self.dev)
        return self._anchors_cache[key]

    def _prepare_filters(self):
        """Per step: data-gradient filters (and the Winograd images of the 3x3 filters) of the CURRENT weights."""
        if BATCHED_LAUNCHES:  # every layer's data-gradient filter in one launch (50 launches of ~6 us each otherwise)
            if getattr(self, "_tbatch", None) is None:
                self._tbatch = T.TransposeBatch([(ly.w, ly.scale, ly.wt, ly.rows, ly.k, ly.k, ly.cin) for ly in self.layers.values()], self.dev)
            self._tbatch.run()
        if not BATCHED_LAUNCHES:
            for ly in self.layers.values():
                T.weight_transpose(ly.w, ly.wt, ly.rows, ly.k, ly.k, ly.cin, scale=ly.scale)
        if self._p16 is not None:  # (bf16 step: both filter sets rounded to bf16, nearest even -- what the kernels' loaders round to)
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(_lib.lib().a3d_f32_to_bf16_scaled(self.params.data_ptr(), self._p16.data_ptr(), self.params.numel(), 1.0, st), "a3d_f32_to_bf16_scaled")
            _lib.check(_lib.lib().a3d_f32_to_bf16_scaled(self._wt.data_ptr(), self._wt16.data_ptr(), self._wt.numel(), 1.0, st), "a3d_f32_to_bf16_scaled")
        for ly in self.layers.values():
            if ly.k == 3 and self.prec != 1:  # (the bf16 step runs its 3x3 layers as direct convolutions)
                T.wino_weight_transform(ly.w, ly.U, ly.rows, ly.cin)
                T.wino_weight_transform(ly.wt, ly.Ut, ly.cin, ly.rows)
                if self.prec == "bf16x3":  # split ONCE per step into preallocated planes (ops.conv2d would re-split per launch)
                    for src, name in ((ly.U, "U3"), (ly.Ut, "Ut3")):
                        rows, cols = src.shape[1], src.shape[2]
                        if cols % 32:
                            continue
                        dst = getattr(ly, name)
                        if dst is None:
                            dst = torch.empty((16, cols // 32, 3, rows, 32), device=src.device, dtype=torch.bfloat16)
                            setattr(ly, name, dst)
                        _lib.check(_lib.lib().a3d_split_bf16x3(src.data_ptr(), dst.data_ptr(), 16, rows, cols, torch.cuda.current_stream().cuda_stream),
                                   "a3d_split_bf16x3")

    def _conv(self, x, pk, **kw):
        """A trainable layer's forward / data-gradient launch in the step's precision."""
        return ops.conv2d(x, pk, precision=self.prec, **kw)

    def _wgrad(self, ly: _Layer, x, dy, accumulate=False):
        """A layer's weight (and bias) gradient.  It hangs OFF the backward pass's critical chain -- it needs the layer's input and the
        gradient of its output, and nothing waits for it before the optimiser step -- so (round 5) it runs on a side stream beside the
        data-gradient chain: at the reference's 2 images per GPU every launch of the step is a fraction of a round of the chip, and the
        ~60 weight-gradient launches hide under the chain.  Same kernels in the same order on ONE side stream (the RPN head's five
        accumulating launches stay ordered): the step's gradients keep their bits."""
        side = self._wg_stream
        if side is None:
            return self._wgrad_now(ly, x, dy, accumulate)
        # (host cost matters: the 2-image step is host-bound.  One reusable event per call site of the step, set_stream instead of the
        # `with torch.cuda.stream(...)` context manager -- 65 -> ~25 us per weight gradient on the host)
        main = self._cur_stream  # (the stream this call is made on: the main one, or the RPN head's)
        ev = self._next_event()
        ev.record(main)
        x.record_stream(side)
        dy.record_stream(side)
        torch.cuda.set_stream(side)
        try:
            side.wait_event(ev)
            self._wgrad_now(ly, x, dy, accumulate)
        finally:
            torch.cuda.set_stream(main)

    def _next_event(self):
        """A reusable ordering event (one per call site of the step; a wait captures the record in front of it, so re-recording is safe)."""
        i = self._wg_calls
        self._wg_calls += 1
        if i == len(self._wg_events):
            self._wg_events.append(torch.cuda.Event())
        return self._wg_events[i]

    def _rpn_head_backward(self, names, dheads, t, feats, st, dP=None):
        """Data and weight gradients of the RPN head (filters shared by the five levels: weight gradients accumulate in level order).
        dP given: the serial form -- every level's gradient is added to dP[level] (p6's to p5 through the stride-2 subsample) and None is
        returned.  dP None: the early form -- returns ({level: gradient}, dp6) for the caller to combine."""
        L, rp = self.layers, "proposal_generator.rpn_head."
        out, dp6 = {}, None
        for li, n in enumerate(names):
            dt = self._conv(dheads[li], L[rp + "pred"].bwd(), gate=t[li], out_dtype=st)
            self._wgrad(L[rp + "pred"], t[li], dheads[li], accumulate=li > 0)
            self._wgrad(L[rp + "conv"], feats[n], dt, accumulate=li > 0)
            if n == "p6":
                dp6 = self._conv(dt, L[rp + "conv"].bwd(), wino=False)
                if dP is not None:
                    T.zero_insert2(dp6, dP["p5"].shape[1], dP["p5"].shape[2], out=dP["p5"], accumulate=True)
            elif dP is not None:
                self._conv(dt, L[rp + "conv"].bwd(), res=dP[n], out=dP[n], wino=False)
            else:
                out[n] = self._conv(dt, L[rp + "conv"].bwd(), wino=False)
        return None if dP is not None else (out, dp6)

    def _rpn_head_backward_early(self, names, dheads, t, feats, st):
        """The RPN head's backward needs nothing but its loss gradient, which exists before the proposals are even selected -- and proposal
        selection, NMS, matching, ROI sampling and the pooler are latency-bound launches of a few workgroups that leave the chip empty
        (~1 ms of the 6 ms step at the reference's 2 images per GPU).  So (round 5) it is enqueued on a second stream right behind the RPN
        loss and runs under that window.  Bits: each pyramid level's gradient is the sum of two terms (ROI pooler + RPN head); the serial
        form adds the RPN term onto the pooler's (conv epilogue, residual), this form lets the pooler's backward add onto the RPN term --
        a + b either way; p5's third term (from p6) is added last in both."""
        main, R = self._cur_stream, self._rpn_stream
        ev = self._next_event()
        ev.record(main)
        for ten in list(dheads) + list(t) + [feats[n] for n in names]:
            ten.record_stream(R)
        torch.cuda.set_stream(R)
        self._cur_stream = R
        try:
            R.wait_event(ev)
            out, dp6 = self._rpn_head_backward(names, dheads, t, feats, st)
            done = self._next_event()
            done.record(R)
        finally:
            torch.cuda.set_stream(main)
            self._cur_stream = main
        for ten in list(out.values()) + [dp6]:
            ten.record_stream(main)
        return out, dp6, done

    def _rpn_labels(self, anchors, gt_boxes, gt_classes, samples, seed, B):
        """Ground truth on the device (fixed-size, counts in a device vector: nothing waits for the host) + the RPN's labels: Matcher + random
        sub-sampling (256 per image, <= half positive), both on the device.  Depends on the ground truth and the anchor grid only."""
        s = self.s
        assert max(len(g) for g in gt_boxes) <= s.max_gt
        gtb = torch.zeros(B, s.max_gt, 4)
        gtc = torch.zeros(B, s.max_gt, dtype=torch.int32)
        gcount = torch.zeros(B, dtype=torch.int32)
        for i, (gb, gc) in enumerate(zip(gt_boxes, gt_classes)):
            gtb[i, : len(gb)] = gb.cpu()
            gtc[i, : len(gb)] = gc.cpu().to(torch.int32)
            gcount[i] = len(gb)
        gtb_d, gtc_d, gcount_d = gtb.to(self.dev, non_blocking=True), gtc.to(self.dev, non_blocking=True), gcount.to(self.dev, non_blocking=True)
        midx, lab = T.match_boxes(anchors, gtb_d, gcount_d, thresholds=s.rpn_iou_thresholds, labels=(0, -1, 1), allow_low_quality=True,
                                  shared=True)
        if samples is None:
            labels_d = T.sample_labels(lab, num=s.rpn_batch_per_image, max_pos=int(s.rpn_batch_per_image * s.rpn_positive_fraction), seed=seed)
        else:
            labels_d = samples["anchor_labels"].to(torch.int8).to(self.dev)
        return gtb_d, gtc_d, gcount_d, midx, lab, labels_d

    def _rpn_labels_early(self, H, W, gt_boxes, gt_classes, samples, seed, B):
        """The same on the second stream, enqueued before the backbone's forward pass: three small launches and the ground-truth upload leave
        the step's dependent chain.  The pyramid's map sizes follow from the input size (every stride-2 stage: (h - 1) // 2 + 1)."""
        f = lambda v: (v - 1) // 2 + 1
        hw, (h, w) = [], (f(f(H)), f(f(W)))
        for _ in self.strides:
            hw.append((h, w))
            h, w = f(h), f(w)
        anchors = self._anchors(hw)
        main, R = self._cur_stream, self._rpn_stream
        ev = self._next_event()
        ev.record(main)
        torch.cuda.set_stream(R)
        self._cur_stream = R
        try:
            R.wait_event(ev)
            out = self._rpn_labels(anchors, gt_boxes, gt_classes, samples, seed, B)
            ready = self._next_event()
            ready.record(R)
        finally:
            torch.cuda.set_stream(main)
            self._cur_stream = main
        for ten in out:
            ten.record_stream(main)
        return hw, out + (ready,)

    def _segment_done(self, i: int):
        """Every launch that writes gradient segment i (self.grad_segments) has been enqueued: fold the segment's parked slice reductions
        and hand it to the gradient exchange.  All weight gradients of a step run in call order on ONE stream (the side stream, or the main
        one), so 'behind the last of them on that stream' is behind all of them.  Without a live exchange (world 1, forward_backward called
        on its own) nothing happens here and the one flush at the end of the backward pass folds everything, as before."""
        if not self._xchg_live:
            return
        if self.grad_overlap.endswith("late"):
            self._late.append(i)
            return
        side, main = self._wg_stream, self._cur_stream
        if side is not None:
            torch.cuda.set_stream(side)
        try:
            if getattr(self, "_defer", None) is not None:
                self._defer.flush(slot=1 + i)
        finally:
            if side is not None:
                torch.cuda.set_stream(main)
        self._xchg.segment_ready(i, side if side is not None else main)

    def _wgrad_now(self, ly: _Layer, x, dy, accumulate=False):
        # The slice reductions of the step's weight gradients are folded in ONE launch at the end of the backward pass (self._defer.flush()
        # in forward_backward): 63 reduce launches of ~17 us each were 11 % of the step at the reference's 2 images per GPU.  Layers whose
        # gradient accumulates over several launches (the RPN head over its five levels) keep the per-launch reduce, which orders them.
        shared = ly.name.startswith("proposal_generator.")
        if BATCHED_LAUNCHES and getattr(self, "_defer", None) is None:
            self._defer = T.DeferredReduces(self.dev)
        T.conv_wgrad(x, dy, ly.dw, KH=ly.k, KW=ly.k, stride=ly.stride, pad=ly.pad, scale=ly.scale, accumulate=accumulate, precision=self.wgrad_prec,
                     defer=self._defer if (BATCHED_LAUNCHES and not shared and not accumulate) else None)
        if ly.db is not None:
            T.colsum(dy, ly.db, accumulate=accumulate)

    # ------------------------------------------------------------------------------------------ the step
    def forward_backward(self, frames_u8: torch.Tensor, gt_boxes: Sequence[torch.Tensor], gt_classes: Sequence[torch.Tensor],
                         samples: Optional[dict] = None, exchange: bool = False) -> Tuple[Dict[str, torch.Tensor], dict]:
        """frames_u8 [B,H,W,3] uint8 BGR on the device; per image gt_boxes [G,4] fp32 / gt_classes [G] int64 (CPU or device).
        Fills self.grads; returns ({loss name: 0-d device tensor}, aux with the sampled index sets).
        exchange=True (what `step` and the reference-style loop pass): the data-parallel gradient exchange starts INSIDE the backward pass,
        segment by segment (parallel.GradientExchange), and `optimizer_step` must follow -- until it has, self.grads is in flight."""
        saved, ops.BF16_SPLITK_AUTO = ops.BF16_SPLITK_AUTO, True  # (split-K by batch size: the training step's launches only, ops.py)
        self._cur_stream, self._wg_calls = torch.cuda.current_stream(), 0
        if not hasattr(self, "_wg_events"):
            self._wg_events = []
        self._xchg_live, self._late = False, []
        if exchange and self.grad_overlap != "0":
            if self._xchg is None:
                self._xchg = GradientExchange(self.grads, self.grad_segments, self.pg, self.grad_payload, force=self.grad_overlap.startswith("force"),
                                              widen=False)
            if self._xchg.active:
                self._xchg.begin()
                self._xchg_live = True
        try:
            return self._forward_backward(frames_u8, gt_boxes, gt_classes, samples)
        finally:
            ops.BF16_SPLITK_AUTO = saved

    def _forward_backward(self, frames_u8, gt_boxes, gt_classes, samples):
        s, L, m = self.s, self.layers, self.model
        B, H, W, _ = frames_u8.shape
        # The per-step filter preparation (data-gradient transposes, bf16 copies: three HBM-bound launches over all parameters) does not touch
        # the frozen stem / res2, so it runs on the side stream beside their forward pass; res3's first launch waits for it.
        prepared = None
        if self._wg_stream is not None:
            main, side = self._cur_stream, self._wg_stream
            ev = self._next_event()
            ev.record(main)  # (behind the previous step's optimiser update)
            torch.cuda.set_stream(side)
            self._cur_stream = side
            try:
                side.wait_event(ev)
                self._prepare_filters()
                prepared = self._next_event()
                prepared.record(side)
            finally:
                torch.cuda.set_stream(main)
                self._cur_stream = main
        else:
            self._prepare_filters()
        seed = (self.seed * 1000003 + self.iter) * 4
        labels_early, early_hw = None, None
        if self._rpn_stream is not None:  # ground-truth upload, anchor matching and label sampling: beside the backbone's forward pass
            early_hw, labels_early = self._rpn_labels_early(H, W, gt_boxes, gt_classes, samples, seed, B)
        saved = {}
        relu_outputs = []  # every ReLU output on the trainable path, in forward order (aux: lets a checker reuse the gates)
        with torch.no_grad():
            x4 = ops.preprocess_u8hwc(frames_u8.contiguous(), self.pixel_mean, self.pixel_std)
            x = m.backbone.bottom_up.forward_stage("res2", m.backbone.bottom_up.stem(x4), frozen=True)  # frozen (FREEZE_AT 2): the inference launches
        res = {"res2": x}
        if prepared is not None:
            self._cur_stream.wait_event(prepared)
        # ---- res3..res5 forward, activations kept
        for name, nblk, _mid, _cout in RES_STAGES:
            for i in range(nblk):
                p = f"backbone.bottom_up.{name}.{i}."
                st = self._st  # (bf16 storage: these four tensors per block are the step's big activations)
                sc = self._conv(x, L[p + "shortcut"].fwd(), out_dtype=st) if i == 0 else x
                a = self._conv(x, L[p + "conv1"].fwd(), out_dtype=st)
                b = self._conv(a, L[p + "conv2"].fwd(), out_dtype=st)
                out = self._conv(b, L[p + "conv3"].fwd(), res=sc, out_dtype=st)
                saved[p] = (x, a, b)
                relu_outputs += [a, b, out]
                x = out
            res[name] = x
        # ---- FPN
        prev, feats = {}, {}
        # (bf16 storage, round 4: the lateral sums, the RPN hidden maps, the box head's hidden rows and the gradients flowing back through
        # them are stored as bf16 too -- what torch.autocast keeps of them; the pyramid p2..p6 itself stays fp32: the pooler and the proposal
        # decoder read it, and its gradient is accumulated from two branches)
        st = self._st
        names = ("p2", "p3", "p4", "p5", "p6")
        rp = "proposal_generator.rpn_head."
        tmap, hmap = {}, {}

        def level_heads(n):  # the RPN head on one pyramid level (shared filters; the levels are independent of each other)
            tmap[n] = self._conv(feats[n], L[rp + "conv"].fwd(), out_dtype=st)
            hmap[n] = self._conv(tmap[n], L[rp + "pred"].fwd())

        R = self._rpn_stream
        if R is None:
            prev[5] = self._conv(res["res5"], L["backbone.fpn_lateral5"].fwd(), out_dtype=st)
            feats["p5"] = self._conv(prev[5], L["backbone.fpn_output5"].fwd())
            for l in (4, 3, 2):
                prev[l] = self._conv(res[f"res{l}"], L[f"backbone.fpn_lateral{l}"].fwd(), res=prev[l + 1], res_ups=True, out_dtype=st)
                feats[f"p{l}"] = self._conv(prev[l], L[f"backbone.fpn_output{l}"].fwd())
            feats["p6"] = ops.subsample2(feats["p5"])
            for n in names:
                level_heads(n)
        else:
            # The top-down lateral chain (four small launches) and the finest level stay on the main stream; the output convs and RPN heads of
            # p3..p6 -- each waiting for its own lateral sum only -- run on the second stream beside them (same launches, same bits).
            main = self._cur_stream
            lat_done = {}
            prev[5] = self._conv(res["res5"], L["backbone.fpn_lateral5"].fwd(), out_dtype=st)
            lat_done[5] = self._next_event()
            lat_done[5].record(main)
            for l in (4, 3, 2):
                prev[l] = self._conv(res[f"res{l}"], L[f"backbone.fpn_lateral{l}"].fwd(), res=prev[l + 1], res_ups=True, out_dtype=st)
                if l > 2:
                    lat_done[l] = self._next_event()
                    lat_done[l].record(main)
            for l in (5, 4, 3):
                prev[l].record_stream(R)
            torch.cuda.set_stream(R)
            self._cur_stream = R
            try:
                for l in (5, 4, 3):
                    R.wait_event(lat_done[l])
                    feats[f"p{l}"] = self._conv(prev[l], L[f"backbone.fpn_output{l}"].fwd())
                    level_heads(f"p{l}")
                    if l == 5:
                        feats["p6"] = ops.subsample2(feats["p5"])
                        level_heads("p6")
                coarse_done = self._next_event()
                coarse_done.record(R)
            finally:
                torch.cuda.set_stream(main)
                self._cur_stream = main
            feats["p2"] = self._conv(prev[2], L["backbone.fpn_output2"].fwd())
            level_heads("p2")
            main.wait_event(coarse_done)
            for n in ("p3", "p4", "p5", "p6"):
                for ten in (feats[n], tmap[n], hmap[n]):
                    ten.record_stream(main)
        t = [tmap[n] for n in names]
        heads = [hmap[n] for n in names]
        feat_hw = [tuple(feats[n].shape[1:3]) for n in names]
        anchors = self._anchors(feat_hw)
        if labels_early is None:
            gtb_d, gtc_d, gcount_d, midx, lab, labels_d = self._rpn_labels(anchors, gt_boxes, gt_classes, samples, seed, B)
        else:  # (matched and sampled on the second stream while the backbone ran)
            assert tuple(feat_hw) == tuple(early_hw), (feat_hw, early_hw)
            gtb_d, gtc_d, gcount_d, midx, lab, labels_d, ready = labels_early
            self._cur_stream.wait_event(ready)
        rpn_l, dheads = T.rpn_loss(heads, self.strides, self.cell_anchors, labels_d, midx, gtb_d, A=3, weights=s.rpn_weights,
                                   normalizer=float(s.rpn_batch_per_image * B))
        early = self._rpn_head_backward_early(names, dheads, t, feats, st) if self._rpn_stream is not None else None
        # ---- proposals (no gradient) + ground truth, Matcher, sub-sampling (512 per image, <= a quarter foreground)
        pb, _ps, _lvl, _pos, pcount = ops.rpn_proposals(heads, self.strides, self.cell_anchors, (H, W), pre_topk=s.rpn_pre_topk_train,
                                                        post_topk=s.rpn_post_topk_train, nms_thresh=s.rpn_nms_thresh, min_size=0.0,
                                                        weights=s.rpn_weights, scale_clamp=math.log(1000.0 / 16))
        allb, bcount = T.append_gt_boxes(pb, pcount, gtb_d, gcount_d)
        pmidx, plab = T.match_boxes(allb, gtb_d, gcount_d, thresholds=(s.roi_iou_threshold,), labels=(0, 1), allow_low_quality=False,
                                    box_count=bcount)
        Rs = s.roi_batch_per_image
        if samples is None:
            roi_boxes, roi_gt, roi_cls, roi_index, rcount_d = T.sample_rois(
                allb, bcount, gtb_d, gtc_d, gcount_d, pmidx, plab, num_classes=s.num_classes, num=Rs,
                max_fg=int(Rs * s.roi_positive_fraction), seed=seed + 1)
        else:  # given index sets (parity tests): the same fixed-size tensors, built with host control flow
            roi_index = torch.full((B, Rs), -1, dtype=torch.int32)
            rcount = torch.zeros(B, dtype=torch.int32)
            for i, ix in enumerate(samples["roi_idx"]):
                roi_index[i, : len(ix)] = ix.to(torch.int32)
                rcount[i] = len(ix)
            roi_index, rcount_d = roi_index.to(self.dev), rcount.to(self.dev)
            live = (roi_index >= 0)
            sel = roi_index.clamp(min=0).long()
            roi_boxes = (torch.gather(allb, 1, sel[:, :, None].expand(B, Rs, 4)) * live[:, :, None]).contiguous()
            msel = torch.gather(pmidx.long(), 1, sel)
            roi_gt = (torch.gather(gtb_d, 1, msel[:, :, None].expand(B, Rs, 4)) * live[:, :, None]).contiguous()
            fg = (torch.gather(plab.long(), 1, sel) == 1) & (gcount_d[:, None] > 0)
            roi_cls = torch.where(fg & live, torch.gather(gtc_d.long(), 1, msel), torch.full_like(msel, s.num_classes)).to(torch.int32).contiguous()
        M = B * Rs  # fixed row count: slots past rcount[b] are zero rows that carry neither loss nor gradient
        # ---- box head forward
        pyr = [feats[n] for n in ("p2", "p3", "p4", "p5")]
        scales = [1.0 / FPN_STRIDES[n] for n in ("p2", "p3", "p4", "p5")]
        pooled = ops.roi_align_fpn(pyr, scales, roi_boxes, rcount_d, 7, 0, True, zero=True)
        bh, bp = "roi_heads.box_head.", "roi_heads.box_predictor."
        xrow = pooled.view(M, 1, 1, 49 * 256)
        h1 = self._conv(xrow, L[bh + "fc1"].fwd(), out_dtype=st)
        h2 = self._conv(h1, L[bh + "fc2"].fwd(), out_dtype=st)
        pred = self._conv(h2, L[bp + "pred"].fwd())
        box_l, dpred = T.box_loss(pred.view(M, 32), roi_cls.view(M), roi_boxes.view(M, 4), roi_gt.view(M, 4), num_classes=s.num_classes,
                                  weights=s.box_weights, count=rcount_d, rows_per_image=Rs)
        losses = {"loss_rpn_cls": rpn_l[0], "loss_rpn_loc": rpn_l[1], "loss_cls": box_l[0], "loss_box_reg": box_l[1]}

        # ======================================== backward ========================================
        dpred = dpred.view(M, 1, 1, 32)
        self._wgrad(L[bp + "pred"], h2, dpred)
        dh2 = self._conv(dpred, L[bp + "pred"].bwd(), gate=h2, out_dtype=st)
        self._wgrad(L[bh + "fc2"], h1, dh2)
        dh1 = self._conv(dh2, L[bh + "fc2"].bwd(), gate=h1, out_dtype=st)
        self._wgrad(L[bh + "fc1"], xrow, dh1)
        self._segment_done(0)  # box head + predictor: 13.9 M of the 41 M gradients leave at the very start of the backward pass
        dpooled = self._conv(dh1, L[bh + "fc1"].bwd())  # [M,1,1,12544]
        if early is not None:  # the RPN head's term of every level is there already (second stream): the pooler's backward adds onto it
            dP, dp6, done = early
            self._cur_stream.wait_event(done)
            T.roi_align_fpn_backward([dP[n] for n in ("p2", "p3", "p4", "p5")], scales, roi_boxes, dpooled.view(M, 7, 7, 256), P=7,
                                     sampling_ratio=0, aligned=True, count=rcount_d)
            T.zero_insert2(dp6, dP["p5"].shape[1], dP["p5"].shape[2], out=dP["p5"], accumulate=True)
        else:
            dP = {n: torch.zeros_like(feats[n]) for n in ("p2", "p3", "p4", "p5")}
            T.roi_align_fpn_backward([dP[n] for n in ("p2", "p3", "p4", "p5")], scales, roi_boxes, dpooled.view(M, 7, 7, 256), P=7,
                                     sampling_ratio=0, aligned=True, count=rcount_d)
            # ---- RPN head backward (weights shared by the five levels: gradients accumulate in level order)
            self._rpn_head_backward(names, dheads, t, feats, st, dP=dP)
        # ---- FPN backward (finest level first: the top-down path carries gradient upwards)
        dprev = {}
        for l in (2, 3, 4, 5):
            lo = L[f"backbone.fpn_output{l}"]
            self._wgrad(lo, prev[l], dP[f"p{l}"])
            dprev[l] = self._conv(dP[f"p{l}"], lo.bwd())
            if l > 2:
                T.sumpool2_add(dprev[l - 1], dprev[l])
            self._wgrad(L[f"backbone.fpn_lateral{l}"], res[f"res{l}"], dprev[l])
        # ---- ResNet backward
        dx_up = None  # gradient arriving at a stage output from the stage above (un-gated)
        for name, nblk, _mid, _cout in reversed(RES_STAGES):
            l = int(name[3:])
            st = self._st
            g = self._conv(dprev[l], L[f"backbone.fpn_lateral{l}"].bwd(), res=dx_up, gate=res[name], out_dtype=st)
            for i in reversed(range(nblk)):
                p = f"backbone.bottom_up.{name}.{i}."
                x_in, a, b = saved[p]
                c1, c2, c3 = L[p + "conv1"], L[p + "conv2"], L[p + "conv3"]
                self._wgrad(c3, b, g)
                db_ = self._conv(g, c3.bwd(), gate=b, out_dtype=st)
                self._wgrad(c2, a, db_)
                da_ = self._conv(db_, c2.bwd(), gate=a, out_dtype=st)
                self._wgrad(c1, x_in, da_)
                if i == 0:
                    self._wgrad(L[p + "shortcut"], x_in, g)
                    if name == "res3":
                        break  # res2 is frozen: nothing below needs a gradient
                    low = self._conv(g, L[p + "shortcut"].bwd())
                    low = self._conv(da_, c1.bwd(), res=low, out=low)
                    dx_up = T.zero_insert2(low, x_in.shape[1], x_in.shape[2])
                else:
                    g = self._conv(da_, c1.bwd(), res=g, gate=x_in, out_dtype=st)
            self._segment_done({"res5": 1, "res4": 2, "res3": 3}[name])  # (res5's segment carries the FPN and the RPN head, finished before it)
        if self._wg_stream is not None:
            self._cur_stream.wait_stream(self._wg_stream)  # every weight gradient has been launched and is waited for here
        if getattr(self, "_defer", None) is not None:
            # every parked weight gradient: one reduce launch.  (Flushed in pieces on the side stream, under the rest of the backward pass:
            # measured slower the finer the pieces -- 352 | 347 | 342 | 338 images/s at 2 images per GPU for one | 24 | 12 | 6 layers per
            # flush: the reduce is HBM-bound and takes its bandwidth from the chain it runs beside.)
            self._defer.flush()
        for i in self._late:  # ("late" forms: every segment behind the whole backward pass, on the main stream)
            self._xchg.segment_ready(i, self._cur_stream)
        self._late = []
        relu_outputs += list(t) + [h1.view(M, -1), h2.view(M, -1)]
        aux = dict(relu_outputs=relu_outputs, anchor_labels=labels_d, roi_index=roi_index, roi_count=rcount_d, roi_cls=roi_cls,
                   proposals=(pb, pcount), heads=heads, feats=feats, pred=pred.view(M, 32), roi_boxes=roi_boxes, anchor_match=(midx, lab))
        return losses, aux

    def autograd_anchor(self) -> torch.Tensor:
        """A leaf that requires grad, so that the loss scalars handed out by PlaneRCNN.training_forward can be `.backward()`-ed."""
        if getattr(self, "_anchor", None) is None:
            self._anchor = torch.zeros((), device=self.dev, requires_grad=True)
        return self._anchor

    def optimizer_step(self):
        s = self.s
        g = self.grads
        if self._xchg_live:  # the segments left during the backward pass: wait for the communication stream
            scale, self._xchg_live = self._xchg.finish(), False
            g16 = self._xchg.reduced_bf16  # (bf16 payload over RCCL: the optimiser reads the sum where the collective left it)
            g = g16 if g16 is not None else g
        else:  # ONE collective behind the backward pass: the flat gradient buffer
            scale = allreduce_gradients(self.grads, self.pg, payload=self.grad_payload)
        T.sgd_momentum(self.params, g, self.momentum, lr=lr_at(self.iter, s), momentum=s.momentum, weight_decay=s.weight_decay,
                       grad_scale=scale, first=self.iter == 0)
        self.iter += 1

    def step(self, frames_u8, gt_boxes, gt_classes, samples=None):
        losses, aux = self.forward_backward(frames_u8, gt_boxes, gt_classes, samples, exchange=True)
        self.optimizer_step()
        return losses, aux
