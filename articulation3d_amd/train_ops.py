"""Host-side wrappers over the training-step kernels of include/a3d.h.  Like ops.py: torch tensors
carry device memory only, all arithmetic happens in liba3d_hip.so."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _lib
from .ops import _p, _req, _stream


ROI_BWD_GATHER = os.environ.get("A3D_ROI_BWD_GATHER", "1") != "0"
# workgroups a transposed-read weight-gradient launch aims for (pixel slices = this / the layer's tiles): ONE round of the 256 CUs -- measured 512 / 384 / 256 / 192 / 128: 792 / 787 / 812-822 / 801 / 768 images/s at 16 images (every slice costs the reduce launch a read of the layer's gradient)
WGRAD_TR_WORKGROUPS = int(os.environ.get("A3D_WGRAD_TR_WGS", "256"))
WGRAD_TR_MIN_PIXELS = int(os.environ.get("A3D_WGRAD_TR_MINPX", "256"))  # pixels per slice at least (4 chunks of 64; 512 / 256 / 128: 312 / 318 / 312 images/s at 2 images)


def choose_wgrad_slices(P: int, tiles: int) -> int:
    """Pixel slices of a weight-gradient GEMM: enough workgroups to fill 256 CUs x 4, each slice >= 64 pixels.
    Depends on the layer shape only (deterministic summation order for a given batch size)."""
    want = max(1, 1024 // max(tiles, 1))
    return int(max(1, min(want, P // 64 if P >= 64 else 1, 256)))


class DeferredReduces:
    """The slice reductions of a step's weight-gradient launches, folded in ONE launch (a3d_wgrad_reduce_batch).  conv_wgrad(...,
    defer=this) launches the partial-sum kernel only and parks its descriptor here; `flush()` uploads the table (pinned, asynchronous)
    and reduces every parked layer.  Workspaces are persistent per layer (keyed by the dw pointer), so nothing is freed under a pending
    reduce.  Same sums in the same order as the per-launch reduce: bit-identical gradients."""

    def __init__(self, device):
        self.device = device
        self.ws = {}      # dw.data_ptr() -> persistent workspace tensor
        self.items = []   # WgradDesc copies of this step
        self._tables = {}  # flush slot -> (host bytes, device copy) of the table last uploaded for it

    def workspace(self, dw: torch.Tensor, nbytes: int) -> torch.Tensor:
        t = self.ws.get(dw.data_ptr())
        if t is None or t.numel() * 4 < nbytes:
            t = self.ws[dw.data_ptr()] = torch.empty(nbytes // 4, device=self.device, dtype=torch.float32)
        return t

    def flush(self, slot: int = 0):
        """Reduce everything parked since the last flush, on the current stream.  `slot` names the call site when a step flushes in
        several pieces (the trainer's gradient-exchange segments): each keeps its own uploaded table."""
        n = len(self.items)
        if not n:
            return
        sz = C.sizeof(_lib.WgradDesc)
        arr = (_lib.WgradDesc * n)(*self.items)
        for i in range(n):  # (the reduce reads neither operand: keep the table identical from step to step)
            arr[i].x = arr[i].dy = None
        raw = C.string_at(C.addressof(arr), n * sz)
        host, dev = self._tables.get(slot, (None, None))
        if raw != host:
            # workspaces, gradients and shapes are the same every step (same batch size), so the table is uploaded once; a change
            # (first step, another batch size) drains the stream first -- a pending reduce may still be reading the old table
            torch.cuda.current_stream().synchronize()
            dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
            self._tables[slot] = (raw, dev)
        _lib.check(_lib.lib().a3d_wgrad_reduce_batch(dev.data_ptr(), n, _stream()), "a3d_wgrad_reduce_batch")
        self.items = []


_WGRAD_DESC_CACHE: dict = {}


def conv_wgrad(x: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor, *, KH: int, KW: int, stride: int, pad: int,
               scale: Optional[torch.Tensor] = None, accumulate: bool = False, splitk: Optional[int] = None, precision: int = 0,
               defer: Optional[DeferredReduces] = None) -> torch.Tensor:
    """dw [Cout, KH*KW*Cin] (=/+=) weight gradient of y = conv(x) given dy (NHWC tensors).
    defer: park the slice reduction in a DeferredReduces (one launch for all layers at its flush()); not with accumulate."""
    # (precision 1 only: x / dy may be STORED as bf16 -- a3d_wgrad_desc.io_bf16; the kernel rounds them to bf16 anyway)
    _req(x, x.dtype if (precision == 1 and x.dtype == torch.bfloat16) else torch.float32)
    _req(dy, dy.dtype if (precision == 1 and dy.dtype == torch.bfloat16) else torch.float32)
    _req(dw)
    B, H, W, Cin = x.shape
    B2, Ho, Wo, Cout = dy.shape
    assert B == B2 and dw.numel() == Cout * KH * KW * Cin, (x.shape, dy.shape, dw.shape)
    # (the descriptor's static fields, the kernel form and the slice count are functions of the layer and the batch shape: kept per key, only
    # the tensor pointers are written per launch -- host time matters at 2 images per GPU)
    key = (B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, bool(accumulate), int(precision), x.dtype, dy.dtype, splitk, scale is None, WGRAD_TR_WORKGROUPS, WGRAD_TR_MIN_PIXELS)
    hit = _WGRAD_DESC_CACHE.get(key)
    if hit is None:
        d = _lib.WgradDesc()
        d.x, d.dy, d.scale, d.dw = _p(x), _p(dy), _p(scale), _p(dw)
        d.B, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = B, H, W, Cin, Ho, Wo, Cout
        d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
        d.accumulate = int(accumulate)
        d.precision = int(precision)
        d.io_bf16 = (1 if x.dtype == torch.bfloat16 else 0) | (2 if dy.dtype == torch.bfloat16 else 0)
        d.splitk = 1
        tiles, red = C.c_int(0), C.c_int(0)
        form = _lib.lib().a3d_wgrad_tiles(C.byref(d), C.byref(tiles), C.byref(red))  # the kernel form the library runs this layer on
        if splitk:
            d.splitk = int(splitk)
        elif form > 0:  # transposed-read form (one 512-thread workgroup per CU, 64-pixel chunks): one round of the chip, >= 4 chunks per slice
            d.splitk = int(max(1, min(WGRAD_TR_WORKGROUPS // max(tiles.value, 1), red.value // WGRAD_TR_MIN_PIXELS, 256)))
        else:
            d.splitk = choose_wgrad_slices(B * Ho * Wo, tiles.value)
        nbytes = _lib.lib().a3d_wgrad_workspace_bytes(C.byref(d))
        d.x = d.dy = d.scale = d.dw = None
        if len(_WGRAD_DESC_CACHE) > 4096:
            _WGRAD_DESC_CACHE.clear()
        hit = _WGRAD_DESC_CACHE[key] = (bytes(d), form, nbytes)
    proto, form, nbytes = hit
    d = _lib.WgradDesc.from_buffer_copy(proto)
    d.x, d.dy, d.scale, d.dw = x.data_ptr(), dy.data_ptr(), _p(scale), dw.data_ptr()
    if defer is not None and not accumulate:
        ws = defer.workspace(dw, nbytes)
        d.defer_reduce = 1
    else:
        defer = None
        ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    d.workspace = ws.data_ptr()
    if defer is not None:
        keep = _lib.WgradDesc()
        C.memmove(C.addressof(keep), C.addressof(d), C.sizeof(_lib.WgradDesc))
        defer.items.append(keep)
    from . import ops as _ops

    if _ops.CONV_TIMING is not None:  # tools/train_bench.py's roofline leg: HIP events around the launch, same record as ops.conv2d's
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(_lib.lib().a3d_conv_wgrad_nhwc_f32(C.byref(d), _stream()), "a3d_conv_wgrad_nhwc_f32")
        e1.record()
        fl = 2.0 * B * Ho * Wo * Cout * KH * KW * Cin  # the pixel-reduction GEMM dY^T . X per tap = the layer's forward FLOPs
        label = {0: "conv_wgrad_kernel", 1: f"conv_wgrad_bf16_kernel<false, {d.io_bf16}>", 2: "conv_wgrad_bf16_kernel<true, 0>"}[int(precision)]
        if form > 0:
            label = f"conv_wgrad_tr_kernel<{form}, {d.io_bf16}>"
        _ops.CONV_TIMING.append((label, fl, e0, e1, f"{B}x{H}x{W}x{Cin}->{Cout} k{KH} s{stride} wgrad sk{d.splitk}", fl,
                                 {0: "f32", 1: "bf16", 2: "bf16x6"}[int(precision)], _stream()))
        return dw
    _lib.check(_lib.lib().a3d_conv_wgrad_nhwc_f32(C.byref(d), _stream()), "a3d_conv_wgrad_nhwc_f32")
    return dw


class TransposeBatch:
    """The data-gradient filters of every trainable layer in ONE launch (a3d_weight_transpose_batch).  The table is built once:
    parameters, scales and scratch filters are views of the trainer's flat buffers, whose addresses do not change."""

    def __init__(self, entries, device):
        # entries: (w, scale or None, wt, Cout, KH, KW, Cin)
        n = len(entries)
        arr = (_lib.TransposeItem * n)()
        blk = 0
        for i, (w, sc, wt, Cout, KH, KW, Cin) in enumerate(entries):
            assert _req(w).numel() == Cout * KH * KW * Cin == _req(wt).numel()
            arr[i].w, arr[i].scale, arr[i].wt = w.data_ptr(), _p(sc), wt.data_ptr()
            arr[i].Cout, arr[i].KH, arr[i].KW, arr[i].Cin, arr[i].block0 = Cout, KH, KW, Cin, blk
            blk += ((Cin + 31) // 32) * ((Cout + 31) // 32) * KH * KW
        self.n, self.blocks = n, blk
        self.keep = [t for e in entries for t in e[:3] if t is not None]
        host = torch.empty(n * C.sizeof(_lib.TransposeItem), dtype=torch.uint8)
        C.memmove(host.data_ptr(), C.addressof(arr), host.numel())
        self.table = host.to(device)

    def run(self):
        _lib.check(_lib.lib().a3d_weight_transpose_batch(self.table.data_ptr(), self.n, self.blocks, _stream()), "a3d_weight_transpose_batch")


def weight_transpose(w: torch.Tensor, out: torch.Tensor, Cout: int, KH: int, KW: int, Cin: int,
                     scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert _req(w).numel() == Cout * KH * KW * Cin == _req(out).numel()
    _lib.check(_lib.lib().a3d_weight_transpose(_p(w), _p(scale), _p(out), Cout, KH, KW, Cin, _stream()), "a3d_weight_transpose")
    return out


def wino_weight_transform(w: torch.Tensor, out: torch.Tensor, Cout: int, Cin: int) -> torch.Tensor:
    assert _req(w).numel() == Cout * 9 * Cin and _req(out).numel() == 16 * Cout * Cin
    _lib.check(_lib.lib().a3d_wino_weight_transform(_p(w), _p(out), Cout, Cin, _stream()), "a3d_wino_weight_transform")
    return out


def zero_insert2(x: torch.Tensor, Ho: int, Wo: int, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    B, H, W, Cc = _req(x).shape
    if out is None:
        assert not accumulate
        out = torch.empty((B, Ho, Wo, Cc), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().a3d_zero_insert2_nhwc(_p(x), _p(out), B, H, W, Cc, Ho, Wo, int(accumulate), _stream()), "a3d_zero_insert2_nhwc")
    return out


def sumpool2_add(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    B, H, W, Cc = _req(y).shape
    assert tuple(_req(x).shape) == (B, 2 * H, 2 * W, Cc)
    _lib.check(_lib.lib().a3d_sumpool2_add_nhwc(_p(x), _p(y), B, H, W, Cc, _stream()), "a3d_sumpool2_add_nhwc")
    return y


def colsum(dy: torch.Tensor, out: torch.Tensor, accumulate: bool = False) -> torch.Tensor:
    Cc = dy.shape[-1]
    M = dy.numel() // Cc
    assert _req(out).numel() >= Cc
    ws = torch.empty(_lib.lib().a3d_colsum_workspace_bytes(Cc) // 4, device=dy.device, dtype=torch.float32)
    if dy.dtype == torch.bfloat16:  # (bf16-stored gradients of the bf16 step: exact widening, the same sums)
        _lib.check(_lib.lib().a3d_colsum_bf16(_p(_req(dy, torch.bfloat16)), _p(out), _p(ws), M, Cc, int(accumulate), _stream()), "a3d_colsum_bf16")
        return out
    _lib.check(_lib.lib().a3d_colsum(_p(_req(dy)), _p(out), _p(ws), M, Cc, int(accumulate), _stream()), "a3d_colsum")
    return out


def roi_align_fpn_backward(dfeats: Sequence[torch.Tensor], scales: Sequence[float], boxes: torch.Tensor, dout: torch.Tensor, *,
                           P: int, sampling_ratio: int, aligned: bool, count: Optional[torch.Tensor] = None,
                           row_offset: Optional[torch.Tensor] = None, scatter: bool = False) -> None:
    """dfeats[l] [B,Hf,Wf,C] += gradient of roi_align_fpn with respect to level l.  boxes [B,R,4], dout [rows,P,P,C].
    scatter=True (or A3D_ROI_BWD_GATHER=0): the float-atomics form (run-to-run summation order)."""
    d = _lib.RoiAlignBwdDesc()
    for l, f in enumerate(dfeats):
        _req(f)
        d.dfeat[l], d.Hf[l], d.Wf[l], d.scale[l] = f.data_ptr(), f.shape[1], f.shape[2], float(scales[l])
    d.L, d.C = len(dfeats), dfeats[0].shape[3]
    d.boxes, d.count, d.row_offset = _p(_req(boxes)), _p(count), _p(row_offset)
    d.B, d.R, d.P, d.sampling_ratio, d.aligned = boxes.shape[0], boxes.shape[1], P, sampling_ratio, int(aligned)
    d.dout = _p(_req(dout))
    if ROI_BWD_GATHER and not scatter:  # tile-gather form: no atomics, fixed summation order (csrc/roi_align.hip)
        ws = torch.empty(max(1, _lib.lib().a3d_roi_align_bwd_workspace_bytes(C.byref(d)) // 4), device=boxes.device, dtype=torch.int32)
        rc = _lib.lib().a3d_roi_align_fpn_backward_gather(C.byref(d), ws.data_ptr(), _stream())
        if rc != -3:  # (A3D_ERR_UNSUPPORTED -- P > 7 or C > 256: the scatter form below)
            _lib.check(rc, "a3d_roi_align_fpn_backward_gather")
            return
    _lib.check(_lib.lib().a3d_roi_align_fpn_backward(C.byref(d), _stream()), "a3d_roi_align_fpn_backward")


def match_boxes(boxes: torch.Tensor, gt_boxes: torch.Tensor, gt_count: torch.Tensor, *, thresholds, labels, allow_low_quality: bool,
                shared: bool = False, box_count: Optional[torch.Tensor] = None, return_iou: bool = False):
    """boxes [N,4] (shared=True: the same anchors for every image) or [B,N,4]; gt_boxes [B,Gmax,4]; gt_count [B] int32.
    -> matched_idx int32 [B,N], label int8 [B,N] (, iou [B,N])."""
    _req(boxes), _req(gt_boxes), _req(gt_count, torch.int32)
    B, Gmax = gt_boxes.shape[:2]
    N = boxes.shape[-2]
    d = _lib.MatchDesc()
    d.boxes, d.box_count, d.gt_boxes, d.gt_count = _p(boxes), _p(box_count), _p(gt_boxes), _p(gt_count)
    d.B, d.N, d.Gmax, d.box_batch_stride = B, N, Gmax, 0 if shared else N
    for i, t in enumerate(thresholds):
        d.thresholds[i] = float(t)
    for i, l in enumerate(labels):
        d.labels[i] = int(l)
    d.n_thresholds, d.allow_low_quality = len(thresholds), int(allow_low_quality)
    best = torch.empty((B, Gmax), device=boxes.device, dtype=torch.int32) if allow_low_quality else None
    midx = torch.zeros((B, N), device=boxes.device, dtype=torch.int32)
    lab = torch.full((B, N), -1, device=boxes.device, dtype=torch.int8)
    iou = torch.zeros((B, N), device=boxes.device, dtype=torch.float32) if return_iou else None
    d.gt_best, d.matched_idx, d.label, d.matched_iou = _p(best), _p(midx), _p(lab), _p(iou)
    _lib.check(_lib.lib().a3d_match_boxes(C.byref(d), _stream()), "a3d_match_boxes")
    return (midx, lab, iou) if return_iou else (midx, lab)


def rpn_loss(heads: Sequence[torch.Tensor], strides: Sequence[int], cell_anchors: torch.Tensor, labels: torch.Tensor,
             matched_idx: torch.Tensor, gt_boxes: torch.Tensor, *, A: int, weights, normalizer: float):
    """heads[l] [B,Hf,Wf,CH] -> (loss [2] = (loss_rpn_cls, loss_rpn_loc), dheads list of the same shapes)."""
    d = _lib.RpnLossDesc()
    dheads = []
    ca = cell_anchors.detach().cpu().float()
    for l, h in enumerate(heads):
        _req(h)
        g = torch.empty_like(h)
        dheads.append(g)
        d.head[l], d.dhead[l], d.Hf[l], d.Wf[l], d.stride[l] = h.data_ptr(), g.data_ptr(), h.shape[1], h.shape[2], int(strides[l])
        for a in range(A):
            for k in range(4):
                d.cell_anchors[l][a][k] = float(ca[l, a, k])
    d.B, d.L, d.A, d.CH = heads[0].shape[0], len(heads), A, heads[0].shape[3]
    d.Atotal, d.Gmax = labels.shape[1], gt_boxes.shape[1]
    d.labels, d.matched_idx, d.gt_boxes = _p(_req(labels, torch.int8)), _p(_req(matched_idx, torch.int32)), _p(_req(gt_boxes))
    for k in range(4):
        d.weights[k] = float(weights[k])
    d.normalizer = float(normalizer)
    ws = torch.empty(_lib.lib().a3d_loss_workspace_bytes() // 4, device=labels.device, dtype=torch.float32)
    loss = torch.empty(2, device=labels.device, dtype=torch.float32)
    d.workspace, d.loss = ws.data_ptr(), loss.data_ptr()
    _lib.check(_lib.lib().a3d_rpn_loss(C.byref(d), _stream()), "a3d_rpn_loss")
    return loss, dheads


def sample_labels(labels: torch.Tensor, *, num: int, max_pos: int, seed: int) -> torch.Tensor:
    """labels int8 [B,N] in {-1,0,1} -> the same with all but a random `num` (<= max_pos positives) set to -1."""
    B, N = _req(labels, torch.int8).shape
    out = torch.empty_like(labels)
    _lib.check(_lib.lib().a3d_sample_labels(_p(labels), _p(out), B, N, int(num), int(max_pos), int(seed) & (2 ** 64 - 1), _stream()),
               "a3d_sample_labels")
    return out


def append_gt_boxes(props: torch.Tensor, count: torch.Tensor, gt_boxes: torch.Tensor, gt_count: torch.Tensor):
    """[B,R,4] live proposals + [B,Gmax,4] ground truth -> ([B,R+Gmax,4], count [B] int32)."""
    B, R, _ = _req(props).shape
    Gmax = _req(gt_boxes).shape[1]
    out = torch.empty((B, R + Gmax, 4), device=props.device, dtype=torch.float32)
    cnt = torch.empty((B,), device=props.device, dtype=torch.int32)
    _lib.check(_lib.lib().a3d_append_gt_boxes(_p(props), _p(_req(count, torch.int32)), _p(gt_boxes), _p(_req(gt_count, torch.int32)), _p(out),
                                              _p(cnt), B, R, Gmax, _stream()), "a3d_append_gt_boxes")
    return out, cnt


def sample_rois(boxes: torch.Tensor, box_count: torch.Tensor, gt_boxes: torch.Tensor, gt_classes: torch.Tensor, gt_count: torch.Tensor,
                matched_idx: torch.Tensor, match_label: torch.Tensor, *, num_classes: int, num: int, max_fg: int, seed: int):
    """-> (roi_boxes [B,num,4], roi_gt_boxes [B,num,4], roi_classes [B,num] int32, index [B,num] int32, count [B] int32)."""
    B, N, _ = _req(boxes).shape
    dev = boxes.device
    d = _lib.RoiSampleDesc()
    d.boxes, d.box_count = _p(boxes), _p(_req(box_count, torch.int32))
    d.gt_boxes, d.gt_classes, d.gt_count = _p(_req(gt_boxes)), _p(_req(gt_classes, torch.int32)), _p(_req(gt_count, torch.int32))
    d.matched_idx, d.match_label = _p(_req(matched_idx, torch.int32)), _p(_req(match_label, torch.int8))
    d.B, d.N, d.Gmax, d.num_classes, d.num, d.max_fg = B, N, gt_boxes.shape[1], num_classes, num, max_fg
    d.seed = int(seed) & (2 ** 64 - 1)
    ob = torch.empty((B, num, 4), device=dev, dtype=torch.float32)
    og = torch.empty((B, num, 4), device=dev, dtype=torch.float32)
    oc = torch.empty((B, num), device=dev, dtype=torch.int32)
    oi = torch.empty((B, num), device=dev, dtype=torch.int32)
    on = torch.empty((B,), device=dev, dtype=torch.int32)
    d.out_boxes, d.out_gt_boxes, d.out_classes, d.out_index, d.out_count = _p(ob), _p(og), _p(oc), _p(oi), _p(on)
    _lib.check(_lib.lib().a3d_sample_rois(C.byref(d), _stream()), "a3d_sample_rois")
    return ob, og, oc, oi, on


def box_loss(pred: torch.Tensor, gt_classes: torch.Tensor, boxes: torch.Tensor, gt_boxes: torch.Tensor, *, num_classes: int, weights,
             count: Optional[torch.Tensor] = None, rows_per_image: int = 0):
    """pred [M,pitch] fused predictor rows -> (loss [2] = (loss_cls, loss_box_reg), dpred [M,pitch]).  With `count` [B] the
    rows are `rows_per_image` per image and only the first count[b] of each image are live."""
    M, pitch = _req(pred).shape
    d = _lib.BoxLossDesc()
    if count is not None:
        d.count, d.R = _p(_req(count, torch.int32)), int(rows_per_image)
    dpred = torch.empty_like(pred)
    d.pred, d.dpred = pred.data_ptr(), dpred.data_ptr()
    d.gt_classes, d.boxes, d.gt_boxes = _p(_req(gt_classes, torch.int32)), _p(_req(boxes)), _p(_req(gt_boxes))
    d.M, d.num_classes, d.pitch = M, num_classes, pitch
    for k in range(4):
        d.weights[k] = float(weights[k])
    ws = torch.empty(_lib.lib().a3d_loss_workspace_bytes() // 4, device=pred.device, dtype=torch.float32)
    loss = torch.empty(2, device=pred.device, dtype=torch.float32)
    d.workspace, d.loss = ws.data_ptr(), loss.data_ptr()
    _lib.check(_lib.lib().a3d_box_loss(C.byref(d), _stream()), "a3d_box_loss")
    return loss, dpred


def sgd_momentum(p: torch.Tensor, g: torch.Tensor, buf: torch.Tensor, *, lr: float, momentum: float, weight_decay: float,
                 grad_scale: float = 1.0, first: bool = False) -> None:
    """g: the flat gradient, fp32 -- or bfloat16: the all-reduced bf16 payload as the collective left it (a3d_sgd_momentum_bf16g)."""
    n = _req(p).numel()
    assert g.numel() == n == _req(buf).numel() and g.is_contiguous()
    if g.dtype == torch.bfloat16:
        _lib.check(_lib.lib().a3d_sgd_momentum_bf16g(_p(p), _p(g), _p(buf), n, lr, momentum, weight_decay, grad_scale, int(first), _stream()),
                   "a3d_sgd_momentum_bf16g")
        return
    _req(g)
    _lib.check(_lib.lib().a3d_sgd_momentum(_p(p), _p(g), _p(buf), n, lr, momentum, weight_decay, grad_scale, int(first), _stream()),
               "a3d_sgd_momentum")
