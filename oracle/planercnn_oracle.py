"""CPU oracle: pure-PyTorch (fp32, CPU) restatement of the PlaneRCNN per-frame detection path.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  The product (articulation3d_amd/) never
imports this file.

What each function follows (reference = /root/reference/articulation3d, pkg = articulation3d/):
  * meta-arch glue      pkg/modeling/meta_arch/planercnn.py:125-219
  * ROI heads glue      pkg/modeling/roi_heads/roi_heads.py:118-273
  * plane head          pkg/modeling/roi_heads/plane_head.py:21-89,127-132
  * axis head           pkg/modeling/roi_heads/axis_head.py:21-129,204-211
  * depth head          pkg/modeling/depth_net/depth_head.py:32-102
  * post-process        pkg/modeling/postprocessing.py:11-75
  * mask paste          pkg/layers/mask_ops.py:16-135
  * process / plane LSQ pkg/utils/arti_vis.py:54-149, create_instances :152-194
  * config values       config/config.yaml (line numbers cited inline as cfg:NN)

The backbone (ResNet-50 + FPN), RPN, anchor generator, box-delta decoding, ROIAlign, NMS, the
Fast R-CNN box head/predictor and the Mask R-CNN mask head are detectron2 / torchvision code
that is NOT vendored under /root/reference (setup.py:10 lists both unpinned; README.md:26,41
names detectron2 0.4/0.6 + torchvision 0.8/0.13).  They are restated from the published
algorithms as written in SURVEY.md Appendix A.1-A.10.

PINNING STATUS
  pinned   : mask paste, plane head, axis head, depth head  -- checked against the reference's own
             modules imported in the build container (oracle/make_golden.py; fixtures in
             tests/golden/).
  PARITY UNPINNED: backbone, FPN, RPN, proposal selection, ROIAlign, NMS, box head/predictor, mask
             head -- their source is absent from the reference tree and the reference has no
             tests or golden vectors; Appendix A is the definition of parity for them.
"""
from __future__ import annotations

import ctypes
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    """C half of the oracle (roi_align / nms), built by oracle/Makefile."""
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liba3d_oracle.so")
        if not os.path.exists(path):
            import subprocess

            subprocess.check_call(["make", "-s", "-C", _HERE])
        _LIB = ctypes.CDLL(path)
    return _LIB


# --------------------------------------------------------------------------------------
# configuration (values of config/config.yaml; key path in the comment)
# --------------------------------------------------------------------------------------
@dataclass
class OracleCfg:
    pixel_mean: Tuple[float, ...] = (103.53, 116.28, 123.675)  # MODEL.PIXEL_MEAN cfg:90-93
    pixel_std: Tuple[float, ...] = (1.0, 1.0, 1.0)  # MODEL.PIXEL_STD cfg:94-97
    anchor_sizes: Tuple[float, ...] = (32, 64, 128, 256, 512)  # cfg:48-54
    anchor_ratios: Tuple[float, ...] = (0.5, 1.0, 2.0)  # cfg:42-46
    rpn_pre_topk: int = 1000  # cfg:295
    rpn_post_topk: int = 1000  # cfg:293
    rpn_nms_thresh: float = 0.7  # cfg:291
    rpn_min_size: float = 0.0  # cfg:100
    rpn_weights: Tuple[float, ...] = (1.0, 1.0, 1.0, 1.0)  # cfg:131-135,273
    box_weights: Tuple[float, ...] = (10.0, 10.0, 5.0, 5.0)  # cfg:192-196
    score_thresh: float = 0.7  # MODEL.ROI_HEADS.SCORE_THRESH_TEST cfg:226
    nms_thresh: float = 0.5  # cfg:223
    num_classes: int = 2  # cfg:224
    dets_per_image: int = 100  # TEST.DETECTIONS_PER_IMAGE cfg:357
    box_pool: Tuple[int, int, bool] = (7, 0, True)  # res, ratio, aligned  cfg:204-206
    mask_pool: Tuple[int, int, bool] = (14, 2, False)  # cfg:253-255
    plane_pool: Tuple[int, int, bool] = (14, 0, False)  # cfg:267-269
    axis_pool: Tuple[int, int, bool] = (14, 0, False)  # cfg:167-169
    post_score_thresh: float = 0.1  # planercnn.py:217
    mask_threshold: float = 0.5  # cfg:250
    mask_on: bool = True
    plane_on: bool = True
    axis_on: bool = True
    depth_on: bool = True
    scale_clamp: float = field(default=math.log(1000.0 / 16))
    # torchvision's batched_nms strategy (see batched_nms below): "plain" = one NMS per category on the boxes as they are (the
    # parity definition); "offset" = torchvision >= 0.9's coordinate trick for every call;
    # "tv-gpu" / "tv-cpu" = torchvision's own size rule on that device (trick up to 5000 / 1000 boxes, per-category loop above).
    nms_strategy: str = "plain"


FPN_STRIDES = {"p2": 4, "p3": 8, "p4": 16, "p5": 32, "p6": 64}
RES_STAGES = (("res2", 3, 64, 256, 1), ("res3", 4, 128, 512, 2), ("res4", 6, 256, 1024, 2), ("res5", 3, 512, 2048, 2))


# --------------------------------------------------------------------------------------
# seeded random-init parameters with detectron2 state_dict names (Appendix A.10)
# --------------------------------------------------------------------------------------
def _msra(w):
    torch.nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
    return w


def _xavier(w):
    torch.nn.init.kaiming_uniform_(w, a=1)
    return w


def _torch_conv_default(cout, cin, k):
    m = torch.nn.Conv2d(cin, cout, k)
    return m.weight.detach().clone(), m.bias.detach().clone()


def _torch_linear_default(cout, cin):
    m = torch.nn.Linear(cin, cout)
    return m.weight.detach().clone(), m.bias.detach().clone()


def init_params(seed: int = 2020, calibrate: bool = True) -> Dict[str, torch.Tensor]:
    """Random-init state_dict with the names of `exps/model_final.pth`.

    With the plain detectron2 initialisers every FrozenBN is the identity and the activations of a
    random-init ResNet-50 grow to ~1e4 by res5: RPN deltas then throw every proposal outside the
    image and the detector is degenerate (0 proposals).  calibrate=True therefore sets the
    running_mean / running_var of every (frozen / eval) batch-norm to the statistics of its input
    on one synthetic calibration frame -- what a trained checkpoint's BN buffers hold -- so that the
    activations stay O(1), ~1000 proposals per frame survive NMS, and the folded scale/shift
    epilogues are exercised.  Weights are shared BY VALUE between the oracle and the HIP path.
    """
    g = torch.random.fork_rng()
    g.__enter__()
    try:
        torch.manual_seed(seed)
        P: Dict[str, torch.Tensor] = {}

        def bn(prefix, c, eps_identity):
            P[prefix + ".weight"] = torch.ones(c)
            P[prefix + ".bias"] = torch.zeros(c)
            P[prefix + ".running_mean"] = torch.zeros(c)
            P[prefix + ".running_var"] = torch.ones(c) - eps_identity

        def rconv(prefix, cout, cin, k):
            P[prefix + ".weight"] = _msra(torch.empty(cout, cin, k, k))
            bn(prefix + ".norm", cout, 1e-5)

        bu = "backbone.bottom_up."
        rconv(bu + "stem.conv1", 64, 3, 7)
        cin = 64
        for name, nblk, mid, cout, _stride in RES_STAGES:
            for i in range(nblk):
                p = f"{bu}{name}.{i}."
                if i == 0:
                    rconv(p + "shortcut", cout, cin, 1)
                rconv(p + "conv1", mid, cin, 1)
                rconv(p + "conv2", mid, mid, 3)
                rconv(p + "conv3", cout, mid, 1)
                cin = cout
        for lvl, c in ((2, 256), (3, 512), (4, 1024), (5, 2048)):
            P[f"backbone.fpn_lateral{lvl}.weight"] = _xavier(torch.empty(256, c, 1, 1))
            P[f"backbone.fpn_lateral{lvl}.bias"] = torch.zeros(256)
            P[f"backbone.fpn_output{lvl}.weight"] = _xavier(torch.empty(256, 256, 3, 3))
            P[f"backbone.fpn_output{lvl}.bias"] = torch.zeros(256)
        rp = "proposal_generator.rpn_head."
        P[rp + "conv.weight"] = torch.empty(256, 256, 3, 3).normal_(std=0.01)
        P[rp + "conv.bias"] = torch.zeros(256)
        P[rp + "objectness_logits.weight"] = torch.empty(3, 256, 1, 1).normal_(std=0.01)
        P[rp + "objectness_logits.bias"] = torch.zeros(3)
        P[rp + "anchor_deltas.weight"] = torch.empty(12, 256, 1, 1).normal_(std=0.01)
        P[rp + "anchor_deltas.bias"] = torch.zeros(12)
        rh = "roi_heads."
        P[rh + "box_head.fc1.weight"] = _xavier(torch.empty(1024, 256 * 7 * 7))
        P[rh + "box_head.fc1.bias"] = torch.zeros(1024)
        P[rh + "box_head.fc2.weight"] = _xavier(torch.empty(1024, 1024))
        P[rh + "box_head.fc2.bias"] = torch.zeros(1024)
        P[rh + "box_predictor.cls_score.weight"] = torch.empty(3, 1024).normal_(std=0.01)
        P[rh + "box_predictor.cls_score.bias"] = torch.zeros(3)
        P[rh + "box_predictor.bbox_pred.weight"] = torch.empty(8, 1024).normal_(std=0.001)
        P[rh + "box_predictor.bbox_pred.bias"] = torch.zeros(8)
        for k in range(1, 5):
            P[rh + f"mask_head.mask_fcn{k}.weight"] = _msra(torch.empty(256, 256, 3, 3))
            P[rh + f"mask_head.mask_fcn{k}.bias"] = torch.zeros(256)
        P[rh + "mask_head.deconv.weight"] = _msra(torch.empty(256, 256, 2, 2))
        P[rh + "mask_head.deconv.bias"] = torch.zeros(256)
        P[rh + "mask_head.predictor.weight"] = torch.empty(1, 256, 1, 1).normal_(std=0.001)
        P[rh + "mask_head.predictor.bias"] = torch.zeros(1)
        for k in range(1, 5):
            P[rh + f"plane_head.plane_conv{k}.weight"] = _msra(torch.empty(256, 256, 3, 3))
            P[rh + f"plane_head.plane_conv{k}.bias"] = torch.zeros(256)
        P[rh + "plane_head.plane_fc1.weight"] = _xavier(torch.empty(1024, 256 * 14 * 14))
        P[rh + "plane_head.plane_fc1.bias"] = torch.zeros(1024)
        P[rh + "plane_head.param_pred.weight"], P[rh + "plane_head.param_pred.bias"] = _torch_linear_default(3, 1024)
        for t in ("R", "T"):
            for k in range(1, 5):
                P[rh + f"axis_head.axis_{t}_conv{k}.weight"] = _msra(torch.empty(256, 256, 3, 3))
                P[rh + f"axis_head.axis_{t}_conv{k}.bias"] = torch.zeros(256)
            P[rh + f"axis_head.axis_{t}_fc1.weight"] = _xavier(torch.empty(1024, 256 * 14 * 14))
            P[rh + f"axis_head.axis_{t}_fc1.bias"] = torch.zeros(1024)
        for nm, n in (("rotation", 2), ("offset", 1), ("translation", 2)):
            P[rh + f"axis_head.{nm}.weight"], P[rh + f"axis_head.{nm}.bias"] = _torch_linear_default(n, 1024)
        dh = "depth_head."
        for i in range(1, 6):
            P[dh + f"conv{i}.0.weight"], P[dh + f"conv{i}.0.bias"] = _torch_conv_default(128, 256, 3)
            _bn_default(P, dh + f"conv{i}.1", 128)
        for i, (ci, co) in enumerate(((128, 128), (256, 128), (256, 128), (256, 128), (256, 64)), start=1):
            P[dh + f"deconv{i}.1.weight"], P[dh + f"deconv{i}.1.bias"] = _torch_conv_default(co, ci, 3)
            _bn_default(P, dh + f"deconv{i}.2", co)
        P[dh + "depth_pred.weight"], P[dh + "depth_pred.bias"] = _torch_conv_default(1, 64, 3)
        if calibrate:
            gen = torch.Generator().manual_seed(seed + 3)
            for k in sorted(P):  # non-zero biases so bias epilogues are exercised
                if k.endswith(".bias") and ".norm." not in k and not k.startswith("depth_head.") and P[k].abs().sum() == 0:
                    P[k] = 0.01 * torch.randn(P[k].shape, generator=gen)
            _calibrate_bn(P, seed)
        return P
    finally:
        g.__exit__(None, None, None)


def _bn_default(P, prefix, c):
    P[prefix + ".weight"] = torch.ones(c)
    P[prefix + ".bias"] = torch.zeros(c)
    P[prefix + ".running_mean"] = torch.zeros(c)
    P[prefix + ".running_var"] = torch.ones(c)


@torch.no_grad()
def _calibrate_bn(P, seed):
    """Set every BN's running stats to its input statistics on one calibration frame; the affine
    weight/bias get a seeded spread (0.5..1.5 / +-0.1) so scale and shift are both non-trivial."""
    frame = torch.as_tensor(synthetic_frames(1, seed + 1)[0].transpose(2, 0, 1).astype("float32"))
    x, _ = preprocess([frame], OracleCfg())
    gen = torch.Generator().manual_seed(seed + 2)

    def cal(t, prefix):
        c = t.shape[1]
        P[prefix + ".running_mean"] = t.mean(dim=(0, 2, 3)).clone()
        P[prefix + ".running_var"] = t.var(dim=(0, 2, 3), unbiased=False).clamp_min(1e-6).clone()
        P[prefix + ".weight"] = 0.5 + torch.rand(c, generator=gen)
        P[prefix + ".bias"] = 0.1 * torch.randn(c, generator=gen)

    def rconv(t, prefix, stride=1, pad=0):
        y = F.conv2d(t, P[prefix + ".weight"], None, stride=stride, padding=pad)
        cal(y, prefix + ".norm")
        return _frozen_bn(y, P, prefix + ".norm")

    bu = "backbone.bottom_up."
    x = F.max_pool2d(F.relu_(rconv(x, bu + "stem.conv1", 2, 3)), 3, 2, 1)
    res = {}
    for name, nblk, _mid, _cout, stride in RES_STAGES:
        for i in range(nblk):
            p = f"{bu}{name}.{i}."
            s = stride if i == 0 else 1
            sc = rconv(x, p + "shortcut", s, 0) if i == 0 else x
            y = F.relu_(rconv(x, p + "conv1", s, 0))
            y = F.relu_(rconv(y, p + "conv2", 1, 1))
            y = rconv(y, p + "conv3", 1, 0)
            x = F.relu_(y + sc)
        res[name] = x
    feats = fpn(res, P)
    dh = "depth_head."

    def dbn(t, pre):
        cal(t, pre)
        return F.batch_norm(t, P[pre + ".running_mean"], P[pre + ".running_var"], P[pre + ".weight"], P[pre + ".bias"],
                            training=False, eps=1e-3)

    def conv(i, t):
        t = F.conv2d(t, P[dh + f"conv{i}.0.weight"], P[dh + f"conv{i}.0.bias"], padding=1)
        return F.leaky_relu(dbn(t, dh + f"conv{i}.1"), 0.01)

    def deconv(i, t):
        t = F.interpolate(t, scale_factor=2, mode="nearest")
        t = F.conv2d(t, P[dh + f"deconv{i}.1.weight"], P[dh + f"deconv{i}.1.bias"], padding=1)
        return F.relu(dbn(t, dh + f"deconv{i}.2"))

    t = deconv(1, conv(1, feats["p6"]))
    t = F.interpolate(t, size=feats["p5"].shape[-2:], mode="bilinear", align_corners=False)
    for i, lv in ((2, "p5"), (3, "p4"), (4, "p3"), (5, "p2")):
        t = deconv(i, torch.cat([conv(i, feats[lv]), t], 1))


# --------------------------------------------------------------------------------------
# A.1 preprocess   (planercnn.py:188-196)
# --------------------------------------------------------------------------------------
def preprocess(images_chw: List[torch.Tensor], cfg: OracleCfg, divis: int = 32):
    mean = torch.tensor(cfg.pixel_mean).view(-1, 1, 1)
    std = torch.tensor(cfg.pixel_std).view(-1, 1, 1)
    imgs = [(x.float() - mean) / std for x in images_chw]
    sizes = [(int(x.shape[-2]), int(x.shape[-1])) for x in imgs]
    H = max(s[0] for s in sizes)
    W = max(s[1] for s in sizes)
    H = (H + divis - 1) // divis * divis
    W = (W + divis - 1) // divis * divis
    out = torch.zeros(len(imgs), 3, H, W)
    for i, x in enumerate(imgs):
        out[i, :, : x.shape[-2], : x.shape[-1]] = x
    return out, sizes


# --------------------------------------------------------------------------------------
# A.2 / A.3 ResNet-50 + FPN
# --------------------------------------------------------------------------------------
def _frozen_bn(x, P, prefix):
    return F.batch_norm(
        x, P[prefix + ".running_mean"], P[prefix + ".running_var"], P[prefix + ".weight"], P[prefix + ".bias"],
        training=False, eps=1e-5,
    )


def _rconv(x, P, prefix, stride=1, pad=0):
    return _frozen_bn(F.conv2d(x, P[prefix + ".weight"], None, stride=stride, padding=pad), P, prefix + ".norm")


def resnet50(x, P):
    bu = "backbone.bottom_up."
    x = F.relu_(_rconv(x, P, bu + "stem.conv1", 2, 3))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = {}
    for name, nblk, _mid, _cout, stride in RES_STAGES:
        for i in range(nblk):
            p = f"{bu}{name}.{i}."
            s = stride if i == 0 else 1
            sc = _rconv(x, P, p + "shortcut", s, 0) if i == 0 else x
            y = F.relu_(_rconv(x, P, p + "conv1", s, 0))  # STRIDE_IN_1X1 cfg:125
            y = F.relu_(_rconv(y, P, p + "conv2", 1, 1))
            y = _rconv(y, P, p + "conv3", 1, 0)
            x = F.relu_(y + sc)
        outs[name] = x
    return outs


def fpn(res, P):
    lat = lambda l, t: F.conv2d(t, P[f"backbone.fpn_lateral{l}.weight"], P[f"backbone.fpn_lateral{l}.bias"])
    out = lambda l, t: F.conv2d(t, P[f"backbone.fpn_output{l}.weight"], P[f"backbone.fpn_output{l}.bias"], padding=1)
    feats = {}
    prev = lat(5, res["res5"])
    feats["p5"] = out(5, prev)
    for l in (4, 3, 2):
        td = F.interpolate(prev, scale_factor=2.0, mode="nearest")
        prev = lat(l, res[f"res{l}"]) + td
        feats[f"p{l}"] = out(l, prev)
    feats["p6"] = F.max_pool2d(feats["p5"], kernel_size=1, stride=2, padding=0)
    return feats


def backbone(x, P):
    return fpn(resnet50(x, P), P)


# --------------------------------------------------------------------------------------
# A.4 / A.5 RPN
# --------------------------------------------------------------------------------------
def cell_anchors(size, ratios):
    a = []
    area = float(size) ** 2
    for r in ratios:
        w = math.sqrt(area / r)
        h = r * w
        a.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(a, dtype=torch.float32)


def grid_anchors(Hf, Wf, stride, size, ratios):
    base = cell_anchors(size, ratios)  # A x 4
    sx = torch.arange(0, Wf * stride, stride, dtype=torch.float32)
    sy = torch.arange(0, Hf * stride, stride, dtype=torch.float32)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
    return (shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4)


def apply_deltas(deltas, boxes, weights, scale_clamp):
    """Box2BoxTransform.apply_deltas (A.5). deltas: N x (k*4), boxes: N x 4."""
    deltas = deltas if deltas.dtype == torch.float64 else deltas.float()  # (float64: oracle/exact.py's yardstick run)
    boxes = boxes.to(deltas.dtype)
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = deltas[:, 0::4] / wx
    dy = deltas[:, 1::4] / wy
    dw = deltas[:, 2::4] / ww
    dh = deltas[:, 3::4] / wh
    dw = torch.clamp(dw, max=scale_clamp)
    dh = torch.clamp(dh, max=scale_clamp)
    pcx = dx * widths[:, None] + ctr_x[:, None]
    pcy = dy * heights[:, None] + ctr_y[:, None]
    pw = torch.exp(dw) * widths[:, None]
    ph = torch.exp(dh) * heights[:, None]
    x1 = pcx - 0.5 * pw
    y1 = pcy - 0.5 * ph
    x2 = pcx + 0.5 * pw
    y2 = pcy + 0.5 * ph
    return torch.stack((x1, y1, x2, y2), dim=-1).reshape(deltas.shape)


def rpn_head(feats: Dict[str, torch.Tensor], P):
    rp = "proposal_generator.rpn_head."
    logits, deltas = [], []
    for name in ("p2", "p3", "p4", "p5", "p6"):
        t = F.relu(F.conv2d(feats[name], P[rp + "conv.weight"], P[rp + "conv.bias"], padding=1))
        lg = F.conv2d(t, P[rp + "objectness_logits.weight"], P[rp + "objectness_logits.bias"])
        dl = F.conv2d(t, P[rp + "anchor_deltas.weight"], P[rp + "anchor_deltas.bias"])
        N, A, H, W = lg.shape
        logits.append(lg.permute(0, 2, 3, 1).reshape(N, -1))
        deltas.append(dl.view(N, A, 4, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, 4))
    return logits, deltas


def topk_stable(scores: torch.Tensor, k: int):
    """Descending top-k, ties -> lower index first (the parity definition, A.5)."""
    vals, idx = torch.sort(scores, dim=-1, descending=True, stable=True)
    return vals[..., :k], idx[..., :k]


def nms_sorted(boxes: torch.Tensor, cats: torch.Tensor, thr: float) -> torch.Tensor:
    """boxes already score-descending. returns bool keep mask (A.6)."""
    n = boxes.shape[0]
    keep = np.zeros(n, dtype=np.uint8)
    if n == 0:
        return torch.zeros(0, dtype=torch.bool)
    b = np.ascontiguousarray(boxes.detach().numpy().astype(np.float32))
    c = np.ascontiguousarray(cats.detach().numpy().astype(np.int32))
    _lib().orc_nms_sorted(
        b.ctypes.data_as(ctypes.c_void_p), c.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n), ctypes.c_float(thr),
        keep.ctypes.data_as(ctypes.c_void_p),
    )
    return torch.from_numpy(keep.astype(bool))


def nms_sorted_py(boxes, cats, thr):
    """Pure-python statement of A.6 for small cases (cross-checks the C kernel)."""
    n = len(boxes)
    b = boxes.numpy().astype(np.float32)
    sup = [False] * n
    keep = [False] * n
    for i in range(n):
        if sup[i]:
            continue
        keep[i] = True
        ai = np.float32(b[i, 2] - b[i, 0]) * np.float32(b[i, 3] - b[i, 1])
        for j in range(i + 1, n):
            if sup[j] or int(cats[j]) != int(cats[i]):
                continue
            w = max(np.float32(0), np.float32(min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0])))
            h = max(np.float32(0), np.float32(min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1])))
            inter = np.float32(w * h)
            aj = np.float32(b[j, 2] - b[j, 0]) * np.float32(b[j, 3] - b[j, 1])
            if np.float32(inter / np.float32(np.float32(ai + aj) - inter)) > np.float32(thr):
                sup[j] = True
    return torch.tensor(keep, dtype=torch.bool)


def batched_nms(boxes, scores, cats, thr, strategy: str = "plain"):
    """Returns kept indices in score-descending order (stable).

    [d2-spec / tv-spec] detectron2's `batched_nms` (layers/nms.py, v0.4-v0.6) forwards to `torchvision.ops.boxes.batched_nms`, which
    since torchvision 0.9 picks between two formulations by the number of boxes (`boxes.numel() > 4000` on the CPU, `> 20000` on a
    GPU -> `_batched_nms_vanilla`: one `nms` per category on the boxes as they are; otherwise `_batched_nms_coordinate_trick`):
        max_coordinate = boxes.max(); offsets = idxs.to(boxes) * (max_coordinate + 1); keep = nms(boxes + offsets[:, None], scores, thr)
    Categories then never overlap, and the IoU of two boxes of one category is evaluated on coordinates shifted by up to
    (C - 1) * (max + 1) -- in fp32 the shift costs the coordinates up to 2-3 low bits, so an IoU within ~1e-6 of the threshold can
    fall on the other side.  `strategy` "plain" is the parity definition (what the HIP kernels implement); "offset" restates the
    trick, "tv-gpu" / "tv-cpu" torchvision's size rule.  tests/test_oracle_golden.py reports whether any committed frame's keep
    set differs between the formulations (the reference's proposal stage holds 5 x 1000 candidates: trick on a GPU, loop on the
    CPU; its box stage <= 2000: trick on a GPU)."""
    order = torch.sort(scores, descending=True, stable=True)[1]
    n = boxes.shape[0]
    trick = strategy == "offset" or (strategy == "tv-gpu" and 4 * n <= 20000) or (strategy == "tv-cpu" and 4 * n <= 4000)
    if trick and n:
        max_coordinate = boxes.max()
        offsets = cats.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
        shifted = boxes + offsets[:, None]
        keep = nms_sorted(shifted[order], torch.zeros(n, dtype=torch.int64), thr)
    else:
        keep = nms_sorted(boxes[order], cats[order], thr)
    return order[keep]


def clip_boxes(b, h, w):
    b = b.clone()
    b[..., 0].clamp_(min=0, max=w)
    b[..., 1].clamp_(min=0, max=h)
    b[..., 2].clamp_(min=0, max=w)
    b[..., 3].clamp_(min=0, max=h)
    return b


def rpn_select_level(logits_l, deltas_l, Hf, Wf, stride, size, cfg: OracleCfg, image_size):
    """One (image, level): top-k (stable), decode, clip, validity.  logits_l [HWA], deltas_l [HWA,4].
    -> idx [k] (anchor index), scores [k], boxes [k,4] clipped, valid [k] bool."""
    anc = grid_anchors(Hf, Wf, stride, size, cfg.anchor_ratios)
    k = min(cfg.rpn_pre_topk, logits_l.shape[0])
    sc, idx = topk_stable(logits_l, k)
    b = apply_deltas(deltas_l[idx], anc[idx], cfg.rpn_weights, cfg.scale_clamp)
    fin = torch.isfinite(b).all(1) & torch.isfinite(sc)
    bc = clip_boxes(b, image_size[0], image_size[1])
    ne = ((bc[:, 2] - bc[:, 0]) > cfg.rpn_min_size) & ((bc[:, 3] - bc[:, 1]) > cfg.rpn_min_size)
    return idx, sc, bc, fin & ne


def rpn_select(logits, deltas, feat_hw, image_sizes, cfg: OracleCfg, return_groups=False, sources: Optional[list] = None):
    """find_top_rpn_proposals (A.5) on given head outputs.  logits[l] [N,HWA], deltas[l] [N,HWA,4].
    `sources` (optional list) receives per image the (level [R], anchor index [R]) of every kept proposal."""
    names = ("p2", "p3", "p4", "p5", "p6")
    N = logits[0].shape[0]
    out, groups = [], []
    for n in range(N):
        bs, ss, ls, gl, ai = [], [], [], [], []
        for li, name in enumerate(names[: len(logits)]):
            Hf, Wf = feat_hw[li]
            idx, sc, bc, valid = rpn_select_level(logits[li][n], deltas[li][n], Hf, Wf, FPN_STRIDES[name],
                                                  cfg.anchor_sizes[li], cfg, image_sizes[n])
            gl.append(dict(idx=idx, scores=sc, boxes=bc, valid=valid))
            bs.append(bc[valid])
            ss.append(sc[valid])
            ls.append(torch.full((int(valid.sum()),), li, dtype=torch.int64))
            ai.append(idx[valid])
        b, s_, l = torch.cat(bs), torch.cat(ss), torch.cat(ls)
        keep = batched_nms(b, s_, l, cfg.rpn_nms_thresh, cfg.nms_strategy)[: cfg.rpn_post_topk]
        out.append((b[keep], s_[keep]))
        if sources is not None:
            sources.append((l[keep], torch.cat(ai)[keep]))
        groups.append(gl)
    return (out, groups) if return_groups else out


def rpn_proposals(feats, P, image_sizes, cfg: OracleCfg, sources: Optional[list] = None):
    """-> list per image of (proposal_boxes Rx4, objectness_logits R)."""
    logits, deltas = rpn_head(feats, P)
    names = ("p2", "p3", "p4", "p5", "p6")
    return rpn_select(logits, deltas, [tuple(feats[n].shape[-2:]) for n in names], image_sizes, cfg, sources=sources)


# --------------------------------------------------------------------------------------
# A.7 ROIPooler / ROIAlign
# --------------------------------------------------------------------------------------
def assign_levels(boxes: torch.Tensor, min_level=2, max_level=5, canon_size=224, canon_level=4):
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    sizes = torch.sqrt(area)
    lv = torch.floor(canon_level + torch.log2(sizes / canon_size + 1e-8))
    lv = torch.clamp(lv, min=min_level, max=max_level)
    return lv.to(torch.int64) - min_level


def roi_align(feat: torch.Tensor, rois: torch.Tensor, P: int, scale: float, ratio: int, aligned: bool):
    if feat.dtype == torch.float64:  # the float64 yardstick of oracle/exact.py: same algorithm, double arithmetic
        feat, rois = feat.contiguous(), rois.contiguous().double()
        out = torch.empty(rois.shape[0], feat.shape[1], P, P, dtype=torch.float64)
        if rois.shape[0]:
            _lib().orc_roi_align_nchw_f64(
                ctypes.c_void_p(feat.data_ptr()), *feat.shape, ctypes.c_void_p(rois.data_ptr()), rois.shape[0], P,
                ctypes.c_double(scale), int(ratio), int(bool(aligned)), ctypes.c_void_p(out.data_ptr()))
        return out
    feat = feat.contiguous().float()
    rois = rois.contiguous().float()
    N, C, H, W = feat.shape
    K = rois.shape[0]
    out = torch.empty(K, C, P, P)
    if K:
        _lib().orc_roi_align_nchw(
            ctypes.c_void_p(feat.data_ptr()), N, C, H, W, ctypes.c_void_p(rois.data_ptr()), K, P,
            ctypes.c_float(scale), int(ratio), int(bool(aligned)), ctypes.c_void_p(out.data_ptr()),
        )
    return out


def roi_align_py(feat, rois, P, scale, ratio, aligned):
    """Slow pure-python statement of A.7 (small cases; cross-checks the C kernel)."""
    N, C, H, W = feat.shape
    out = torch.zeros(rois.shape[0], C, P, P)
    f32 = np.float32

    def bil(plane, y, x):
        if y < -1.0 or y > H or x < -1.0 or x > W:
            return torch.zeros(C)
        y = max(y, f32(0))
        x = max(x, f32(0))
        yl, xl = int(y), int(x)
        if yl >= H - 1:
            yh = yl = H - 1
            y = f32(yl)
        else:
            yh = yl + 1
        if xl >= W - 1:
            xh = xl = W - 1
            x = f32(xl)
        else:
            xh = xl + 1
        ly, lx = f32(y - f32(yl)), f32(x - f32(xl))
        hy, hx = f32(1) - ly, f32(1) - lx
        return (f32(hy * hx) * plane[:, yl, xl] + f32(hy * lx) * plane[:, yl, xh]
                + f32(ly * hx) * plane[:, yh, xl] + f32(ly * lx) * plane[:, yh, xh])

    for k in range(rois.shape[0]):
        r = rois[k].numpy().astype(np.float32)
        b = int(r[0])
        off = f32(0.5) if aligned else f32(0)
        x1, y1, x2, y2 = (f32(r[1] * f32(scale)) - off, f32(r[2] * f32(scale)) - off,
                          f32(r[3] * f32(scale)) - off, f32(r[4] * f32(scale)) - off)
        rw, rh = f32(x2 - x1), f32(y2 - y1)
        if not aligned:
            rw, rh = max(rw, f32(1)), max(rh, f32(1))
        bh, bw = f32(rh / f32(P)), f32(rw / f32(P))
        gh = ratio if ratio > 0 else int(math.ceil(f32(rh / f32(P))))
        gw = ratio if ratio > 0 else int(math.ceil(f32(rw / f32(P))))
        cnt = f32(max(gh * gw, 1))
        for ph in range(P):
            for pw in range(P):
                acc = torch.zeros(C)
                for iy in range(gh):
                    y = f32(f32(y1 + f32(f32(ph) * bh)) + f32(f32(f32(iy) + f32(0.5)) * bh) / f32(gh))
                    for ix in range(gw):
                        x = f32(f32(x1 + f32(f32(pw) * bw)) + f32(f32(f32(ix) + f32(0.5)) * bw) / f32(gw))
                        acc = acc + bil(feat[b], y, x)
                out[k, :, ph, pw] = acc / cnt
    return out


def roi_pool_fpn(feats: Dict[str, torch.Tensor], box_lists: List[torch.Tensor], P: int, ratio: int, aligned: bool):
    """ROIPooler over (p2..p5).  box_lists: per-image Ki x 4 -> sum(Ki) x C x P x P."""
    names = ("p2", "p3", "p4", "p5")
    rois = torch.cat([torch.cat((torch.full((len(b), 1), float(i)), b.float()), 1) for i, b in enumerate(box_lists)], 0)
    C = feats["p2"].shape[1]
    out = torch.zeros(rois.shape[0], C, P, P)
    if rois.shape[0] == 0:
        return out
    lv = assign_levels(rois[:, 1:])
    for li, name in enumerate(names):
        sel = (lv == li).nonzero().squeeze(1)
        if len(sel):
            out[sel] = roi_align(feats[name], rois[sel], P, 1.0 / FPN_STRIDES[name], ratio, aligned)
    return out


# --------------------------------------------------------------------------------------
# A.8 box head + predictor
# --------------------------------------------------------------------------------------
def box_head(x, P):
    rh = "roi_heads.box_head."
    x = torch.flatten(x, 1)
    x = F.relu(F.linear(x, P[rh + "fc1.weight"], P[rh + "fc1.bias"]))
    x = F.relu(F.linear(x, P[rh + "fc2.weight"], P[rh + "fc2.bias"]))
    return x


def box_predictor(x, P):
    rh = "roi_heads.box_predictor."
    return (F.linear(x, P[rh + "cls_score.weight"], P[rh + "cls_score.bias"]),
            F.linear(x, P[rh + "bbox_pred.weight"], P[rh + "bbox_pred.bias"]))


def fast_rcnn_inference_single(boxes, scores, image_size, cfg: OracleCfg):
    """boxes R x (C*4) decoded, scores R x (C+1) softmaxed.  -> pred_boxes, scores, classes, kept row idx."""
    valid = torch.isfinite(boxes).all(1) & torch.isfinite(scores).all(1)
    rows = torch.arange(boxes.shape[0])
    if not valid.all():
        boxes, scores, rows = boxes[valid], scores[valid], rows[valid]
    scores = scores[:, :-1]
    C = boxes.shape[1] // 4
    boxes = clip_boxes(boxes.reshape(-1, 4), image_size[0], image_size[1]).view(-1, C, 4)
    fmask = scores > cfg.score_thresh
    finds = fmask.nonzero()
    b = boxes[fmask]
    s = scores[fmask]
    keep = batched_nms(b, s, finds[:, 1], cfg.nms_thresh, cfg.nms_strategy)
    if cfg.dets_per_image >= 0:
        keep = keep[: cfg.dets_per_image]
    return b[keep], s[keep], finds[keep, 1], rows[finds[keep, 0]]


def box_inference(feats, proposals, P, image_sizes, cfg: OracleCfg):
    pooled = roi_pool_fpn(feats, [p[0] for p in proposals], *cfg.box_pool)
    x = box_head(pooled, P)
    cls, dlt = box_predictor(x, P)
    nper = [len(p[0]) for p in proposals]
    allb = torch.cat([p[0] for p in proposals], 0)
    dec = apply_deltas(dlt, allb, cfg.box_weights, cfg.scale_clamp)
    probs = F.softmax(cls, dim=-1)
    res = []
    for bx, pr, sz in zip(dec.split(nper), probs.split(nper), image_sizes):
        res.append(fast_rcnn_inference_single(bx, pr, sz, cfg))
    return res, dict(pooled=pooled, fc=x, cls=cls, deltas=dlt)


# --------------------------------------------------------------------------------------
# A.9 mask head ; reference plane / axis / depth heads
# --------------------------------------------------------------------------------------
def mask_head(x, P):
    rh = "roi_heads.mask_head."
    for k in range(1, 5):
        x = F.relu(F.conv2d(x, P[rh + f"mask_fcn{k}.weight"], P[rh + f"mask_fcn{k}.bias"], padding=1))
    x = F.relu(F.conv_transpose2d(x, P[rh + "deconv.weight"], P[rh + "deconv.bias"], stride=2))
    x = F.conv2d(x, P[rh + "predictor.weight"], P[rh + "predictor.bias"])
    return x.sigmoid()  # (D,1,28,28), class agnostic


def plane_head(x, P, return_raw=False):
    """plane_head.py:71-82.  return_raw: also the param_pred output before F.normalize (:80) -- the quantity whose rounding error
    the normalisation amplifies by 1 / |raw| (tests/test_gpu_e2e.py compares it directly)."""
    rh = "roi_heads.plane_head."
    for k in range(1, 5):
        x = F.relu(F.conv2d(x, P[rh + f"plane_conv{k}.weight"], P[rh + f"plane_conv{k}.bias"], padding=1))
    x = torch.flatten(x, 1)
    x = F.relu(F.linear(x, P[rh + "plane_fc1.weight"], P[rh + "plane_fc1.bias"]))
    x = F.linear(x, P[rh + "param_pred.weight"], P[rh + "param_pred.bias"])
    return (F.normalize(x, p=2, dim=1), x) if return_raw else F.normalize(x, p=2, dim=1)


def axis_head(x, P, return_raw=False):
    """axis_head.py:95-120.  return_raw: also (rotation | offset, translation) before F.normalize (:106,:120)."""
    rh = "roi_heads.axis_head."

    def tower(t, x):
        for k in range(1, 5):
            x = F.relu(F.conv2d(x, P[rh + f"axis_{t}_conv{k}.weight"], P[rh + f"axis_{t}_conv{k}.bias"], padding=1))
        x = torch.flatten(x, 1)
        return F.relu(F.linear(x, P[rh + f"axis_{t}_fc1.weight"], P[rh + f"axis_{t}_fc1.bias"]))

    xr = tower("R", x)
    rot_raw = F.linear(xr, P[rh + "rotation.weight"], P[rh + "rotation.bias"])
    rot = F.normalize(rot_raw, p=2, dim=1)
    off = F.linear(xr, P[rh + "offset.weight"], P[rh + "offset.bias"])
    xt = tower("T", x)
    tran_raw = F.linear(xt, P[rh + "translation.weight"], P[rh + "translation.bias"])
    tran = F.normalize(tran_raw, p=2, dim=1)
    if return_raw:
        return torch.cat((rot, off), 1), tran, torch.cat((rot_raw, off), 1), tran_raw
    return torch.cat((rot, off), 1), tran


def depth_head(feats, P):
    """depth_head.py:72-89"""
    dh = "depth_head."

    def bn(x, pre):
        return F.batch_norm(x, P[pre + ".running_mean"], P[pre + ".running_var"], P[pre + ".weight"], P[pre + ".bias"],
                            training=False, eps=1e-3)

    def conv(i, x):
        x = F.conv2d(x, P[dh + f"conv{i}.0.weight"], P[dh + f"conv{i}.0.bias"], padding=1)
        return F.leaky_relu(bn(x, dh + f"conv{i}.1"), 0.01)

    def deconv(i, x):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = F.conv2d(x, P[dh + f"deconv{i}.1.weight"], P[dh + f"deconv{i}.1.bias"], padding=1)
        return F.relu(bn(x, dh + f"deconv{i}.2"))

    x = deconv(1, conv(1, feats["p6"]))
    x = F.interpolate(x, size=feats["p5"].shape[-2:], mode="bilinear", align_corners=False)
    x = deconv(2, torch.cat([conv(2, feats["p5"]), x], 1))
    x = deconv(3, torch.cat([conv(3, feats["p4"]), x], 1))
    x = deconv(4, torch.cat([conv(4, feats["p3"]), x], 1))
    x = deconv(5, torch.cat([conv(5, feats["p2"]), x], 1))
    x = F.conv2d(x, P[dh + "depth_pred.weight"], P[dh + "depth_pred.bias"], padding=1)
    H, W = x.shape[-2] * 2, x.shape[-1] * 2
    x = F.interpolate(x, size=(H, W), mode="bilinear", align_corners=False)
    return x.view(-1, H, W)


# --------------------------------------------------------------------------------------
# post-process: postprocessing.py:11-75, mask_ops.py:16-135
# --------------------------------------------------------------------------------------
def paste_masks(masks: torch.Tensor, boxes: torch.Tensor, img_h: int, img_w: int, threshold: float = 0.5):
    """masks D x M x M probs, boxes D x 4 -> D x H x W bool.  Whole-image (GPU-path) form of
    mask_ops._do_paste_mask with skip_empty=False (:41-60); the per-box CPU path (:35-40) samples the
    same grid on a sub-window, so both give the same values inside it and zeros outside."""
    D = masks.shape[0]
    if D == 0:
        return torch.zeros(0, img_h, img_w, dtype=torch.bool)
    x0, y0, x1, y1 = torch.split(boxes, 1, dim=1)
    img_y = torch.arange(0, img_h, dtype=torch.float32) + 0.5
    img_x = torch.arange(0, img_w, dtype=torch.float32) + 0.5
    img_y = (img_y - y0) / (y1 - y0) * 2 - 1
    img_x = (img_x - x0) / (x1 - x0) * 2 - 1
    gx = img_x[:, None, :].expand(D, img_h, img_w)
    gy = img_y[:, :, None].expand(D, img_h, img_w)
    grid = torch.stack([gx, gy], dim=3)
    out = F.grid_sample(masks[:, None].float(), grid, align_corners=False)
    return out[:, 0] >= threshold


def detector_postprocess(det: dict, out_h: int, out_w: int, cfg: OracleCfg):
    """det: dict(pred_boxes Dx4, scores D, pred_classes D, [pred_masks Dx1x28x28, pred_plane, ...],
    image_size (h, w)).  Returns a new dict."""
    ih, iw = det["image_size"]
    sx, sy = out_w / iw, out_h / ih
    sel = det["scores"] >= cfg.post_score_thresh
    r = {k: (v[sel] if torch.is_tensor(v) else v) for k, v in det.items()}
    b = r["pred_boxes"].clone()
    b[:, 0::2] *= sx
    b[:, 1::2] *= sy
    b = clip_boxes(b, out_h, out_w)
    ne = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
    r = {k: (v[ne] if torch.is_tensor(v) else v) for k, v in r.items()}
    r["pred_boxes"] = b[ne]
    r["image_size"] = (out_h, out_w)
    if "pred_masks" in r:
        r["pred_masks"] = paste_masks(r["pred_masks"][:, 0], r["pred_boxes"], out_h, out_w, cfg.mask_threshold)
    return r


# --------------------------------------------------------------------------------------
# PlaneRCNN_Branch.process: arti_vis.py:63-149
# --------------------------------------------------------------------------------------
def k_inv_dot_xy1(h=480, w=640, focal=571.623718):
    """arti_vis.py:101-123 (float64 then FloatTensor, :50)."""
    K = np.array([[focal, 0, 319.5], [0, focal, 239.5], [0, 0, 1]], dtype=np.float64)
    Kinv = np.linalg.inv(K)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float64) / h * 480, np.arange(w, dtype=np.float64) / w * 640, indexing="ij")
    pts = np.stack([xs, ys, np.ones_like(xs)], 0).reshape(3, -1)
    rays = (Kinv @ pts).reshape(3, h, w)
    return torch.from_numpy(rays).float()


def override_depth(depth: torch.Tensor, masks: torch.Tensor, planes: torch.Tensor, rays: torch.Tensor):
    """arti_vis.py:90-99,125-149.  depth HxW, masks DxHxW bool, planes Dx3 (unit normals) -> Dx3."""
    xyz = (rays * depth).numpy()  # :98
    pl = planes.clone().numpy().astype(np.float32)
    pl[:, [1, 2]] = pl[:, [2, 1]]  # :130
    pl[:, 1] = -pl[:, 1]  # :131
    out = []
    for m, p in zip(masks.numpy().astype(bool), pl):
        if m.sum() == 0:  # :136-138
            out.append(p)
            continue
        pts = xyz[:, m]
        offset = np.linalg.norm(p)
        normal = p / max(offset, 1e-8)
        off_new = (normal @ pts).mean()
        out.append(normal * off_new)
    if len(out) == 0:
        return planes
    o = torch.from_numpy(np.stack(out).astype(np.float32))
    o[:, [1, 2]] = o[:, [2, 1]]  # :146
    o[:, 2] = -o[:, 2]  # :147
    return o


# --------------------------------------------------------------------------------------
# the whole per-frame path: planercnn.py:148-184 + arti_vis.py:54-87
# --------------------------------------------------------------------------------------
@torch.no_grad()
def detect(images_chw: List[torch.Tensor], P, cfg: Optional[OracleCfg] = None, given_boxes=None, return_aux=False, features=None):
    """images: list of CHW float32 BGR 0-255.  Returns list[dict] (one per image) with
    pred_boxes, scores, pred_classes, pred_masks (bool HxW), pred_plane, pred_rot_axis,
    pred_tran_axis, depth, plus 'plane_offset' = process()'s overridden planes.
    `features` (optional) replaces the backbone output: the stability search (oracle/seed_search.py) re-runs everything
    behind the backbone on perturbed feature maps."""
    cfg = cfg or OracleCfg()
    x, sizes = preprocess(images_chw, cfg)
    feats = backbone(x, P) if features is None else features
    aux = {"features": feats}
    if given_boxes is None:
        src = []
        props = rpn_proposals(feats, P, sizes, cfg, sources=src)
        dets, box_aux = box_inference(feats, props, P, sizes, cfg)
        aux.update(proposals=props, box=box_aux, proposal_sources=src)
    else:  # forward_with_given_boxes entry, roi_heads.py:147
        dets = [(b.float(), torch.ones(len(b)), torch.zeros(len(b), dtype=torch.int64), torch.arange(len(b))) for b in given_boxes]
    depth = depth_head(feats, P) if cfg.depth_on else [None] * len(sizes)
    boxes = [d[0] for d in dets]
    nper = [len(b) for b in boxes]
    # (prop_rows: the proposal each detection came from -- bookkeeping for oracle/exact.py, not a reference field)
    results = [dict(pred_boxes=d[0], scores=d[1], pred_classes=d[2], prop_rows=d[3], image_size=sz) for d, sz in zip(dets, sizes)]
    if cfg.mask_on:
        m = mask_head(roi_pool_fpn(feats, boxes, *cfg.mask_pool), P)
        for r, mm in zip(results, m.split(nper)):
            r["pred_masks"] = mm
    if cfg.plane_on:
        pl, plr = plane_head(roi_pool_fpn(feats, boxes, *cfg.plane_pool), P, return_raw=True)
        for r, pp, pr in zip(results, pl.split(nper), plr.split(nper)):
            r["pred_plane"] = pp
            r["raw_plane"] = pr  # (checker bookkeeping, not a reference field: the vector before F.normalize)
    if cfg.axis_on:
        ra, ta, rar, tar = axis_head(roi_pool_fpn(feats, boxes, *cfg.axis_pool), P, return_raw=True)
        for r, a, t, ar, tr in zip(results, ra.split(nper), ta.split(nper), rar.split(nper), tar.split(nper)):
            r["pred_rot_axis"] = a
            r["pred_tran_axis"] = t
            r["raw_rot"], r["raw_tran"] = ar, tr
    rays = k_inv_dot_xy1()
    outs = []
    for i, (r, sz) in enumerate(zip(results, sizes)):
        r = detector_postprocess(r, sz[0], sz[1], cfg)
        r["depth"] = depth[i]
        if cfg.depth_on and cfg.plane_on and cfg.mask_on and tuple(sz) == (480, 640):
            r["plane_offset"] = override_depth(depth[i], r["pred_masks"], r["pred_plane"], rays)
        outs.append(r)
    return (outs, aux) if return_aux else outs


def synthetic_frames(n: int, seed: int = 2020, h: int = 480, w: int = 640) -> np.ndarray:
    """uint8 uniform[0,255] BGR frames (n,h,w,3), SURVEY.md 8d."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)


def frames_to_chw(frames: np.ndarray) -> List[torch.Tensor]:
    """arti_vis.py:58"""
    return [torch.as_tensor(f.transpose(2, 0, 1).astype("float32")) for f in frames]
