"""Parameter holders for the layers of the detection path.

Each holder keeps its parameters under the SAME state_dict names as the detectron2 / reference modules
(so `exps/model_final.pth`-style checkpoints load unchanged) and lowers itself
once, lazily, to the packed layout of the HIP implicit-GEMM kernel (ops.PackedConv).  forward() works on
fp32 NHWC device tensors and only launches kernels through `ops`.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
from torch import nn

from .. import ops
from ..ops import ACT_LEAKY, ACT_NONE, ACT_RELU  # noqa: F401


def c2_msra_fill(weight: torch.Tensor, bias: Optional[torch.Tensor] = None):
    nn.init.kaiming_normal_(weight, mode="fan_out", nonlinearity="relu")
    if bias is not None:
        nn.init.constant_(bias, 0)


def c2_xavier_fill(weight: torch.Tensor, bias: Optional[torch.Tensor] = None):
    nn.init.kaiming_uniform_(weight, a=1)
    if bias is not None:
        nn.init.constant_(bias, 0)


def to_nhwc(t: torch.Tensor) -> torch.Tensor:
    """NCHW-shaped tensor -> contiguous NHWC.  Zero-copy when `t` is a channels-last view (what this
    package's modules hand to each other)."""
    return t.permute(0, 2, 3, 1).contiguous()


def to_nchw_view(t: torch.Tensor) -> torch.Tensor:
    """NHWC buffer -> NCHW-shaped view (channels_last strides), the shape the reference's callers see."""
    return t.permute(0, 3, 1, 2)


class _CalibrationState:
    """While `active`, every conv that carries a (frozen / eval) batch-norm first measures the statistics
    of its own un-normalised output on the batch flowing through and stores them in the norm's running
    buffers (what a trained checkpoint holds).  Used once, offline, by utils.synthetic.calibrate_batchnorm to
    give random-init weights well-conditioned activations; never active during inference."""

    active = False


def _calibrate_norm(norm, y_raw: torch.Tensor, eps_floor: float = 1e-6):
    flat = y_raw.reshape(-1, y_raw.shape[-1])[:, : norm.running_mean.numel()]
    norm.running_mean.copy_(flat.mean(0))
    norm.running_var.copy_(flat.var(0, unbiased=False).clamp_min(eps_floor))


def _publish():
    """A freshly packed weight set is computed on the CURRENT stream but read afterwards by launches on any stream (1-2 frame batches
    run one RPN head layer on three streams, the three ROI heads side by side): drain the stream once before the cache is used."""
    if torch.cuda.is_available() and not os.environ.get("A3D_NO_PUBLISH"):
        torch.cuda.current_stream().synchronize()


class _Packable(nn.Module):
    """Caches the packed weights; re-packs when a parameter/buffer was modified or moved."""

    pin_precision = None  # 2: the precision audit (PlaneRCNN.audit_precision) pinned this layer to bf16x3; applied to every (re)pack
    b2b_second = False    # True: second layer of a back-to-back pointwise pair (backbone.ResNet marks them): bf16x3 in every form
    _a3d_name = ""        # qualified module name, filled in by the audit for its report

    def __init__(self):
        super().__init__()
        self._pack_cache = None
        self._pack_key = None
        self._key_src = None

    # The key is asked for on EVERY forward call (a 1-frame pass makes ~100 of them): walking parameters() / buffers() through the module
    # machinery cost ~14 us per call -- 1.2 ms of host time per frame in the reference's per-frame loop, which is host-bound.  The (owning
    # dict, name) slots are therefore found once and only looked up afterwards: a tensor replaced in its slot (load_state_dict copies in
    # place, .to() / .cuda() replace buffers and parameter data) is seen through the lookup; anything that changes the SET of slots
    # (attribute assignment on this module, _apply) drops the slot list; a slot added on a CHILD module later (child.register_buffer,
    # add_module, a replaced grandchild -- none of which pass through this module's __setattr__) is caught by a structural fingerprint kept
    # beside the list: the slot counts and the child identities of every module walked, compared on each call (~0.5 us for a conv + norm).
    def __setattr__(self, name, value):
        if name not in ("_pack_cache", "_pack_key", "_key_src"):
            self.__dict__["_key_src"] = None
        super().__setattr__(name, value)

    def _apply(self, fn, recurse=True):
        self.__dict__["_key_src"] = None
        return super()._apply(fn, recurse)

    @staticmethod
    def _shape_of(mods):
        n, kids = 0, []
        for m in mods:
            n += len(m._parameters) + len(m._buffers)
            kids += [id(c) for c in m._modules.values()]
        return n, tuple(kids)

    def _key(self):
        cached = self.__dict__.get("_key_src")
        if cached is not None:
            src, mods, shape = cached
            if shape != self._shape_of(mods):
                cached = None
        if cached is None:
            src, mods = [], list(self.modules())
            for m in mods:
                src += [(m._parameters, k) for k in m._parameters] + [(m._buffers, k) for k in m._buffers]
            shape = self._shape_of(mods)
            self.__dict__["_key_src"] = (src, mods, shape)
        out = []
        for d, k in src:
            t = d.get(k)
            if t is not None:
                out.append((t.data_ptr(), t._version, t.device))
        return tuple(out)

    def packed(self) -> ops.PackedConv:
        key = self._key()
        if self._pack_cache is None or key != self._pack_key:
            self._pack_cache = self._pack()
            self._pack_key = key
            _publish()
        pin = self.pin_precision
        if pin is None and self.b2b_second and ops.B2B_FUSED:
            # the squeeze behind a block output that the previous block's launch can produce (ops.conv2d_b2b): that launch forms it in the
            # bf16x3 arithmetic, so the layer runs bf16x3 in EVERY form -- also as a launch of its own (batches under one round of the
            # chip) -- and a frame's bits do not depend on how it was batched
            pin = 2
        self._pack_cache.pin_precision, self._pack_cache.name = pin, self._a3d_name
        return self._pack_cache

    def _pack(self):
        raise NotImplementedError


class FrozenBatchNorm2d(nn.Module):
    """Buffers of detectron2's FrozenBatchNorm2d (eps 1e-5); folded into the conv epilogue."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)


class Conv2d(_Packable):
    """conv (+ optional bias) (+ optional FrozenBN as `.norm`) (+ activation), NHWC in / NHWC out."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, norm=None, act=ACT_NONE):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.act = kernel_size, stride, padding, act
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.norm = norm
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def _pack(self):
        bn = None
        if self.norm is not None:
            n = self.norm
            bn = (n.weight, n.bias, n.running_mean, n.running_var, n.eps)
        if self.in_channels == 3 and self.kernel_size == 7:
            return ops.pack_stem(self.weight, bn, device=self.weight.device)
        return ops.pack_conv(self.weight, self.bias, bn, self.stride, self.padding, self.act, device=self.weight.device)

    def forward(self, x, **kw):
        if _CalibrationState.active and self.norm is not None:
            stem = self.in_channels == 3 and self.kernel_size == 7
            ident = (torch.ones_like(self.norm.weight), torch.zeros_like(self.norm.bias), torch.zeros_like(self.norm.running_mean),
                     torch.ones_like(self.norm.running_var) - self.norm.eps, self.norm.eps)
            raw = ops.pack_stem(self.weight, ident, device=self.weight.device) if stem else \
                ops.pack_conv(self.weight, self.bias, ident, self.stride, self.padding, ACT_NONE, device=self.weight.device)
            _calibrate_norm(self.norm, ops.conv2d(x, raw, act=ACT_NONE, **{k: v for k, v in kw.items() if k in ("x2", "ups")}))
        return ops.conv2d(x, self.packed(), **kw)


class BNConv2d(_Packable):
    """nn.Sequential(Conv2d, BatchNorm2d, act) of the depth head (depth_head.py:32-46): parameters live in
    the parent under `<name>.0.*` / `<name>.1.*` (or `.1` / `.2` for the deconv form)."""

    def __init__(self, conv: nn.Conv2d, bn: nn.BatchNorm2d, act: int):
        super().__init__()
        self.conv, self.bn, self.act = conv, bn, act

    def _key(self):
        ts = list(self.conv.parameters()) + list(self.bn.parameters()) + list(self.bn.buffers())
        return tuple((t.data_ptr(), t._version, str(t.device)) for t in ts)

    def _pack(self):
        bn = (self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var, self.bn.eps)
        return ops.pack_conv(self.conv.weight, self.conv.bias, bn, 1, 1, self.act, device=self.conv.weight.device)

    def forward(self, x, **kw):
        if _CalibrationState.active:
            raw = ops.pack_conv(self.conv.weight, self.conv.bias, None, 1, 1, ACT_NONE, device=self.conv.weight.device)
            _calibrate_norm(self.bn, ops.conv2d(x, raw, **kw))
        if kw.get("ups"):  # conv over a nearest-x2 upsampled input: four 2x2 source-grid convs (4/9 of the FLOPs)
            return ops.conv2d_ups(x, self.packed_phases(), x2=kw.get("x2"))
        return ops.conv2d(x, self.packed(), **kw)

    def packed_phases(self):
        key = self._key()
        if getattr(self, "_phase_cache", None) is None or key != self._phase_key:
            bn = (self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var, self.bn.eps)
            self._phase_cache = ops.pack_conv_ups_phases(self.conv.weight, self.conv.bias, bn, self.act, device=self.conv.weight.device)
            self._phase_key = key
            _publish()
        for q in self._phase_cache:
            q.pin_precision, q.name = self.pin_precision, self._a3d_name
        return self._phase_cache


class Linear(_Packable):
    """nn.Linear holder.  `chw` = (C,H,W) of the flattened NCHW feature this layer consumes in the reference;
    the packed copy is re-ordered for the NHWC activations used here."""

    def __init__(self, in_features, out_features, chw=None, act=ACT_NONE):
        super().__init__()
        self.in_features, self.out_features, self.chw, self.act = in_features, out_features, chw, act
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)

    def _pack(self):
        return ops.pack_linear(self.weight, self.bias, self.chw, self.act, device=self.weight.device)

    def forward(self, x, **kw):
        return ops.linear(x, self.packed(), **kw)
