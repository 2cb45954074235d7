"""Thin host-side wrappers over the C-ABI kernels (include/a3d.h).

torch is used for device memory and streams only: every function takes / returns torch CUDA tensors,
hands their `data_ptr()` and the current HIP stream to liba3d_hip.so and does no arithmetic itself.
All activations are fp32 NHWC.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ACT_LEAKY, ACT_NONE, ACT_RELU  # noqa: F401

GROUP_CAP = 1024
fptr_t = C.c_void_p

# Optional per-launch timing of the conv-GEMM kernel (bench.py's roofline leg): when a list is installed
# here, conv2d() brackets its launch with HIP events on the launch stream and appends
# (kernel variant as recorded by the dispatcher, algorithmic_flops, start_event, end_event, shape, executed_flops, pipe).
# executed_flops = fp32 multiply-adds the kernel's contraction performs x2 (Winograd F(2x2,3x3): 16 per 2x2 output tile and channel
# pair instead of 36; phase convs of an upsampled 3x3: 4 taps instead of 9); pipe = "f32" (fp32-input MFMA), "bf16" (one bf16 MFMA
# per product) or "bf16x6" (bf16x3 mode: SIX bf16 MFMA products per fp32 multiply-add).
CONV_TIMING: Optional[list] = None

# Arithmetic of conv2d() calls that do not ask for one (a3d_conv_desc.precision):
#   3 = "fp16x2", THE DEFAULT (round 2, late): fp32-grade arithmetic on the 16-bit matrix pipe with THREE MFMAs per k step.  Every
#       fp32 operand, scaled by a power of two s, is split in two fp16 terms (x * s = h + l: h carries 11 significant bits, l the
#       next 11; an fp16 x fp16 product is exact in fp32) and h.h + h.l + l.h accumulate in fp32; the dropped l.l is <= 2^-22
#       relative.  fp16 has 5 exponent bits, hence the scale: s puts the largest magnitude of the operand in [2^14, 2^15) -- per
#       IMAGE (or ROI) for activations, taken from maxima that the producing kernel's epilogue records (`_a3d_amax`, amax_of below),
#       so a frame's result never depends on the rest of its batch; once per layer for the filter.  Values down to 2^-18 of the
#       image's maximum keep the full 22 bits, smaller ones an absolute 2^-40 of it.  Measured error against float64 is BELOW
#       bf16x3's and the fp32 MFMA's on every layer shape, also with images 10^6 apart in magnitude in one batch
#       (tools/x3_bench.py, tools/h2_check.py; tests test_fp16x2_*), and the whole GPU suite passes under it.
#       Kernels: the bf16x3 ones with two operand planes (template flag F16 of conv_bf16x3.hip, conv_bf16x3_wide.hip and the wide
#       Winograd GEMM of conv_wino.hip); 3x3 layers whose channel count the wide Winograd tiles do not fit run the direct form.
#   2 = "bf16x3" (A3D_PRECISION=2 / bench.py --precision bf16x3), the default of round 2 until fp16x2: every fp32 operand split
#       EXACTLY into three bf16 terms (no scale needed: bf16 has fp32's exponent range), six MFMAs per k step (hi.hi, hi.mid,
#       mid.hi, mid.mid, hi.lo, lo.hi); the dropped terms are <= 2^-24 relative.  Error against float64 at or below the fp32
#       MFMA's (test_bf16x3_kernel_is_fp32_grade, test_winograd_split_operand_gemm_is_fp32_grade).
#   0 = fp32-input MFMA (A3D_PRECISION=0 / bench.py --precision fp32): the round-1 default, bit-compatible with it; the
#       one-launch Winograd kernel (csrc/conv_wino_fused.hip) belongs to this mode.  gfx950's fp32-input MFMA runs at the fp32
#       VECTOR rate (157 TFLOP/s, 1/16 of the 16-bit pipes) and shares that datapath with the VALU.
#   1 = bf16 MFMA with fp32 accumulation on plain convolutions / linears: autocast-level error, opt-in, never a parity mode.
DEFAULT_PRECISION = int(os.environ.get("A3D_PRECISION", "3"))


def H3_KINDS(p, wino_ok: bool, splitk: int) -> bool:
    """Layer kinds that have an fp16x2 kernel (the others keep bf16x3 inside mode 3)."""
    return True  # (Winograd layers whose channel count the wide kernels' 128-wide tiles do not fit -- res2's 64 -> 64 -- run the direct form)


def last_conv_variant() -> str:
    """Kernel instantiation dispatched by the calling thread's last conv launch (a3d_last_conv_variant, include/a3d.h):
    the dispatcher's own record, e.g. "conv_pw_kernel<2,2,16> 128x128 persistent"."""
    return _lib.lib().a3d_last_conv_variant().decode()


# The raw handle of the calling thread's current stream, straight from the binding layer: `torch.cuda.current_stream()` builds a Stream object
# through four layers of device-index helpers (~6 us), and every launch asks -- a third of the host time of the 2-images-per-GPU training
# step, which is host-bound (tools/probes/train_host_profile.py).
_raw_stream, _raw_device = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("articulation3d_amd ops need CUDA/HIP tensors (there is no CPU fallback)")
    if t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError(f"expected contiguous {dtype} tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


# --------------------------------------------------------------------------------------------------
# packed weights
# --------------------------------------------------------------------------------------------------
@dataclass
class PackedConv:
    w: torch.Tensor  # [cols, Kpad]
    scale: Optional[torch.Tensor]
    shift: Optional[torch.Tensor]
    KH: int
    KW: int
    stride: int
    pad: int
    Cin: int  # total input channels (both sources)
    cols: int  # GEMM columns (multiple of 4)
    Kpad: int
    act: int = ACT_NONE
    pixshuf: bool = False
    stem: bool = False
    w_wino: Optional[torch.Tensor] = None  # [16, cols, Cin] Winograd-domain weights (3x3 s1 p1 convs)
    w_wino_x3: Optional[torch.Tensor] = None  # [16, Cin/32, 3, cols, 32] bf16 planes of w_wino, made at the first precision-2 use
    w_wino_cm: Optional[torch.Tensor] = None  # [Cin/8, 16, cols, 8] chunk-major copy of w_wino: the one-launch Winograd kernel
    phase: int = 0  # 1..4: one output phase of a conv over a nearest-x2 upsampled input (see a3d_conv_desc.phase)
    presplit: bool = False  # set by the pack_* functions (module-cached weights): precision-2 launches may cache w_x3 below
    w_x3: Optional[torch.Tensor] = None  # [Kpad/16, 3, cols, 16] bf16 planes of w (a3d_conv_desc.w_x3), made at the first such use
    w_wino_h2: Optional[torch.Tensor] = None  # [16, Cin/32, 2, cols, 32] fp16 planes of w_wino * its scale (precision 3)
    w_h2: Optional[torch.Tensor] = None  # [Kpad/16, 2, cols, 16] fp16 planes of w * w_scale (a3d_conv_desc.w_x3 at precision 3)
    w_b16: Optional[torch.Tensor] = None  # [cols, Kpad] bf16 = w rounded to nearest even (a3d_conv_desc.w_bf16; the trainer refreshes it once per step)
    pin_precision: Optional[int] = None  # 2: this layer runs bf16x3 (exact splits, no window) whatever the mode -- set by the precision audit
    name: str = ""  # owner module's qualified name, when known (audit reports)

    @property
    def out_channels(self) -> int:
        return self.cols // 4 if self.pixshuf else self.cols


def _pad_rows(t: torch.Tensor, mult: int = 4) -> torch.Tensor:
    n = t.shape[0]
    n_pad = (n + mult - 1) // mult * mult
    if n_pad == n:
        return t
    out = t.new_zeros((n_pad,) + tuple(t.shape[1:]))
    out[:n] = t
    return out


def fold_bn(weight, bias, mean, var, eps, conv_bias=None):
    scale = weight * torch.rsqrt(var + eps)
    shift = bias - mean * scale
    if conv_bias is not None:
        shift = shift + conv_bias * scale
    return scale, shift


def pack_conv(weight: torch.Tensor, bias=None, bn=None, stride=1, pad=0, act=ACT_NONE, device="cuda") -> PackedConv:
    """weight [Cout, Cin, KH, KW] (torch layout) -> [Cout_pad4, KH*KW*Cin], k = (kh, kw, c)."""
    Cout, Cin, KH, KW = weight.shape
    if Cin % 32:
        raise ValueError("generic conv path needs Cin % 32 == 0")
    w = weight.detach().float().permute(0, 2, 3, 1).reshape(Cout, KH * KW * Cin)
    scale = shift = None
    if bn is not None:
        scale, shift = fold_bn(*[t.detach().float() for t in bn[:4]], bn[4], None if bias is None else bias.detach().float())
    elif bias is not None:
        shift = bias.detach().float()
    w = _pad_rows(w)
    cols = w.shape[0]
    if scale is not None:
        scale = _pad_rows(scale)
    if shift is not None:
        shift = _pad_rows(shift)
    dev = lambda t: None if t is None else t.contiguous().to(device)
    pk = PackedConv(dev(w), dev(scale), dev(shift), KH, KW, stride, pad, Cin, cols, KH * KW * Cin, act, presplit=True)
    if KH == 3 and KW == 3 and stride == 1 and pad == 1 and Cin % 16 == 0:
        U = winograd_weights(_pad_rows(weight.detach().float()))
        pk.w_wino = dev(U)
        pk.w_wino_cm = dev(winograd_weights_chunk_major(U))
    return pk


def pack_conv_ups_phases(weight: torch.Tensor, bias=None, bn=None, act=ACT_NONE, device="cuda") -> List[PackedConv]:
    """3x3 pad-1 conv applied to a nearest-x2 upsampled input == four 2x2 convs on the source grid, one per output
    phase (dy,dx), whose taps are sums of the original taps that land on the same source pixel:
    dy=0: rows {kh0 | kh1+kh2}, dy=1: rows {kh0+kh1 | kh2}; same for columns.  4/9 of the FLOPs."""
    Cout, Cin, KH, KW = weight.shape
    assert KH == 3 and KW == 3 and Cin % 32 == 0
    w = weight.detach().double()
    groups = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}
    scale = shift = None
    if bn is not None:
        scale, shift = fold_bn(*[t.detach().float() for t in bn[:4]], bn[4], None if bias is None else bias.detach().float())
    elif bias is not None:
        shift = bias.detach().float()
    dev = lambda t: None if t is None else t.contiguous().to(device)
    scale_d, shift_d = dev(None if scale is None else _pad_rows(scale)), dev(None if shift is None else _pad_rows(shift))
    out = []
    for dy in (0, 1):
        for dx in (0, 1):
            wp = w.new_zeros(Cout, Cin, 2, 2)
            for a, khs in enumerate(groups[dy]):
                for b, kws in enumerate(groups[dx]):
                    for kh in khs:
                        for kw in kws:
                            wp[:, :, a, b] += w[:, :, kh, kw]
            wk = _pad_rows(wp.float().permute(0, 2, 3, 1).reshape(Cout, 4 * Cin))
            out.append(PackedConv(dev(wk), scale_d, shift_d, 2, 2, 1, 0, Cin, wk.shape[0], 4 * Cin, act, phase=1 + dy * 2 + dx, presplit=True))
    # ONE power-of-two filter scale for the layer, not one per phase: the fused four-phase launch splits all four filters under the scale
    # of their common maximum, and a phase launched alone must split ITS filter under the same scale to produce the same bits (the
    # pre-summed taps of the phases differ in magnitude; under another scale a weight below 2^-18 of the maximum lands in fp16's
    # subnormal range in one form only -- found by tools/fuzz_kernels.py as a 1-ulp difference on one layer in 640)
    common = _pow2_scale_host(max(float(q.w.abs().max()) for q in out))
    for q in out:
        q._w_scale = common
    return out


def pack_conv_ups_fused(phases: Sequence[PackedConv]) -> Optional[PackedConv]:
    """The four phase filters of pack_conv_ups_phases as ONE 3x3 source-grid convolution with 4 x Cout columns (a3d_conv_desc.phase
    == 5): column 128 g + 32 p + c = phase p of output channel 32 g + c, whose 2x2 taps sit at rows kh - dy, columns kw - dx of
    the 3x3 neighbourhood; scale / shift replicated in the same column order.  None when the channel count does not fit (32 | Cout)."""
    p0 = phases[0]
    Cout, CinT = p0.cols, p0.Cin
    if Cout % 32 or CinT % 32 or any(q.cols != Cout or q.Cin != CinT or q.phase != i + 1 for i, q in enumerate(phases)):
        return None
    dev = p0.w.device
    w = torch.zeros((4 * Cout, 3, 3, CinT), device=dev, dtype=torch.float32)
    co = torch.arange(Cout, device=dev)
    for ph, q in enumerate(phases):
        dy, dx = ph >> 1, ph & 1
        col = (co // 32) * 128 + 32 * ph + (co % 32)
        w[col, dy:dy + 2, dx:dx + 2, :] = q.w[:Cout].view(Cout, 2, 2, CinT)
    col_of = torch.empty(4 * Cout, dtype=torch.int64, device=dev)  # real channel of every GEMM column
    j = torch.arange(4 * Cout, device=dev)
    col_of = (j // 128) * 32 + (j % 32)
    scale = None if p0.scale is None else p0.scale[col_of].contiguous()
    shift = None if p0.shift is None else p0.shift[col_of].contiguous()
    return PackedConv(w.reshape(4 * Cout, 9 * CinT).contiguous(), scale, shift, 3, 3, 1, 1, CinT, 4 * Cout, 9 * CinT, p0.act, phase=5, presplit=True)


_WINO_G = ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0))


def winograd_weights(weight: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 3, 3] -> U = G g G^T as [16, Cout, Cin] (frequency f = 4u + v), the layout of a3d_conv_desc.w_wino.
    Weight preparation only (once per layer at pack time)."""
    G = torch.tensor(_WINO_G, dtype=torch.float64, device=weight.device)
    U = torch.einsum("up,ncpq,vq->uvnc", G, weight.double(), G)
    return U.reshape(16, weight.shape[0], weight.shape[1]).float().contiguous()


def winograd_weights_chunk_major(U: torch.Tensor) -> torch.Tensor:
    """U [16, Cout, C] -> a3d_conv_desc.w_wino_cm, the one-launch Winograd kernel's weight image:
    [C/8][16 planes][ceil(Cout/64) tiles][k half 2][channel 64][4], i.e. element (chunk c, plane f, output channel n = 64 t + r,
    input channel k = 8 c + 4 h + j).  One (chunk, plane, tile, k half) is a contiguous 1 KiB run that the kernel moves to LDS
    verbatim with one LDS-DMA instruction; Cout is zero-padded to a multiple of 64."""
    f, n, c = U.shape
    nt = (n + 63) // 64
    Up = U.new_zeros((f, nt * 64, c))
    Up[:, :n] = U
    #        f   t   r  c/8   h  j
    v = Up.view(f, nt, 64, c // 8, 2, 4)
    return v.permute(3, 0, 1, 4, 2, 5).contiguous()


def pack_stem(weight: torch.Tensor, bn, device="cuda") -> PackedConv:
    """7x7 s2 p3 stem on NHWC4 input: [64,3,7,7] -> [64][7][8][4] (kw 7 and channel 3 are zero)."""
    Cout = weight.shape[0]
    w = weight.new_zeros(Cout, 7, 8, 4)
    w[:, :, :7, :3] = weight.detach().float().permute(0, 2, 3, 1)
    scale, shift = fold_bn(*[t.detach().float() for t in bn[:4]], bn[4])
    return PackedConv(w.reshape(Cout, 224).contiguous().to(device), scale.contiguous().to(device),
                      shift.contiguous().to(device), 7, 7, 2, 3, 4, Cout, 224, ACT_RELU, stem=True, presplit=True)


def pack_linear(weight: torch.Tensor, bias=None, chw: Optional[Tuple[int, int, int]] = None, act=ACT_NONE,
                device="cuda") -> PackedConv:
    """nn.Linear weight [N, K].  chw=(C,H,W): the reference flattens NCHW (plane_head.py:76) while the
    pooled / conv activations here are NHWC, so K is re-ordered (c,h,w) -> (h,w,c) once at load."""
    N, K = weight.shape
    w = weight.detach().float()
    if chw is not None:
        c, h, ww = chw
        w = w.view(N, c, h, ww).permute(0, 2, 3, 1).reshape(N, K)
    if K % 32:
        raise ValueError("K % 32 != 0")
    w = _pad_rows(w)
    shift = None if bias is None else _pad_rows(bias.detach().float())
    dev = lambda t: None if t is None else t.contiguous().to(device)
    return PackedConv(dev(w), None, dev(shift), 1, 1, 1, 0, K, w.shape[0], K, act, presplit=True)


def pack_deconv2x2(weight: torch.Tensor, bias, act=ACT_RELU, device="cuda") -> PackedConv:
    """ConvTranspose2d(k=2, s=2) weight [Cin, Cout, 2, 2] -> GEMM columns (dy, dx, co)."""
    Cin, Cout = weight.shape[:2]
    w = weight.detach().float().permute(2, 3, 1, 0).reshape(4 * Cout, Cin)  # [(dy,dx,co), ci]
    shift = bias.detach().float().repeat(4)
    return PackedConv(w.contiguous().to(device), None, shift.contiguous().to(device), 1, 1, 1, 0, Cin, 4 * Cout, Cin,
                      act, pixshuf=True, presplit=True)


def pack_fused_rows(weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], device="cuda") -> PackedConv:
    w = torch.cat([x.detach().float().reshape(x.shape[0], -1) for x in weights], 0)
    b = torch.cat([x.detach().float() for x in biases], 0)
    K = w.shape[1]
    w = _pad_rows(w)
    b = _pad_rows(b)
    return PackedConv(w.contiguous().to(device), None, b.contiguous().to(device), 1, 1, 1, 0, K, w.shape[0], K, ACT_NONE, presplit=True)


# --------------------------------------------------------------------------------------------------
# kernels
# --------------------------------------------------------------------------------------------------
# ---- per-image maxima of activation tensors (a3d_conv_desc.in_amax / y_amax): the power-of-two scales of the fp16x2 split ------
# A tensor [B, ...] produced by a split-operand launch carries `_a3d_amax`, a device tensor [B] with max |t[b]| (recorded by the
# launch's epilogue into a zero-initialised slot).  Views made on the hot path hand it on with keep_amax(); any other tensor gets its
# maxima from a3d_absmax_rows on first use.  Slots come from zeroed arena chunks (one fill per ~256k floats, freed with their tensors).
WINO_PLANE_SPLIT = os.environ.get("A3D_WINO_PLANE_SPLIT", "1") != "0"  # (False: small Winograd problems keep the one-launch form; same bits)
_AMAX_CHUNK = 1 << 18
AMAX_MISSES: Optional[list] = None  # debugging: install a list to record the tensors whose maxima had to be computed by a3d_absmax_rows
_amax_arena: dict = {}


def amax_reserve(n: int, device) -> None:
    """Make sure the current chunk has room for `n` more floats, allocating (and zero-filling, on the CURRENT stream) a new one if
    not.  Called at the start of a batch, before any branch forks onto a side stream: every later slot of the batch then comes from
    a chunk whose fill is ordered before all of the batch's launches on every stream."""
    a = _amax_arena.get(device)
    if a is None or a[1] + n > a[0].numel():
        _amax_arena[device] = [torch.zeros(max(_AMAX_CHUNK, n), device=device, dtype=torch.float32), 0]


def amax_slot(n: int, device) -> torch.Tensor:
    a = _amax_arena.get(device)
    if a is None or a[1] + n > a[0].numel():
        amax_reserve(n, device)
        a = _amax_arena[device]
        if not os.environ.get("A3D_NO_PUBLISH"):
            torch.cuda.current_stream().synchronize()  # (an unreserved refill: its fill must be visible to every stream)
    t = a[0][a[1]:a[1] + n]
    a[1] += (n + 3) // 4 * 4
    return t


def amax_of(x: torch.Tensor) -> torch.Tensor:
    a = getattr(x, "_a3d_amax", None)
    if a is not None and a.numel() == x.shape[0] and a.device == x.device:
        return a
    if AMAX_MISSES is not None:
        AMAX_MISSES.append(tuple(x.shape))
    a = amax_slot(x.shape[0], x.device)
    _lib.check(_lib.lib().a3d_absmax_rows(x.data_ptr(), a.data_ptr(), x.shape[0], x.numel() // x.shape[0], _stream()), "a3d_absmax_rows")
    x._a3d_amax = a
    return a


def keep_amax(new: torch.Tensor, old: torch.Tensor) -> torch.Tensor:
    """`new` is a view / reshape of `old` with the same leading dimension: it keeps old's recorded maxima."""
    a = getattr(old, "_a3d_amax", None)
    if a is not None and new.shape[0] == old.shape[0]:
        new._a3d_amax = a
    return new


def _const_amax(t: torch.Tensor, bound: float) -> None:
    if DEFAULT_PRECISION == 3:
        t._a3d_amax = torch.full((t.shape[0],), float(bound), device=t.device, dtype=torch.float32)


def _pow2_scale_host(amax: float) -> float:
    return 1.0 if not (amax > 0.0) else 2.0 ** (14 - math.frexp(amax)[1] + 1)


_WINO_SHARE: Optional[dict] = None
WINO_SHARE_ENABLED = os.environ.get("A3D_WINO_SHARE", "1") != "0"  # schedule-only switch (tests compare both settings bit for bit)


class share_wino_input:
    """Scope in which the Winograd input transform of each listed tensor is computed once and reused by every 3x3 layer
    that reads it (the RPN conv and the depth head's lateral conv both read the same FPN level; the plane and axis heads
    read the same pooled ROI tensor).  V = B^T d B depends on the input only, so the layers' results are bit-identical to
    the unshared form; the saving is one wino_input_kernel launch (a full read of the level + a 4x-sized write) per reuse.
    The V tensors live until the scope exits -- exit it only after every stream that consumed them has been joined."""

    def __init__(self, tensors: Sequence[torch.Tensor]):
        self.map = {t.data_ptr(): [t, None] for t in tensors} if WINO_SHARE_ENABLED else {}

    def __enter__(self):
        global _WINO_SHARE
        self.prev, _WINO_SHARE = _WINO_SHARE, self.map
        return self

    def __exit__(self, *exc):
        global _WINO_SHARE
        _WINO_SHARE = self.prev
        self.map = {}
        return False


_ADDR_LIMIT = (1 << 32) - 1  # the kernels address every operand with 32-bit buffer offsets: one launch sees < 4 GiB per tensor
WINO_LEVELS = os.environ.get("A3D_WINO_LEVELS", "1") != "0"  # schedule-only switch (False: one launch per map; same bits)


def conv2d_levels(xs: Sequence[torch.Tensor], p: PackedConv) -> list:
    """The SAME 3x3 s1 p1 layer applied to several maps (the RPN head's conv over the pyramid levels: StandardRPNHead, planercnn.py:168;
    SURVEY K5) as ONE Winograd GEMM launch over the concatenated tiles (a3d_wino_gemm_levels): every map is transformed into its slice
    of one V buffer, the GEMM walks all tiles and looks each tile's map up in its epilogue.  The partial rounds of the small maps vanish
    (p2-p6 at 64 frames: 19 + 5 + 2 + 1 + 1 rounds of the chip -> 25) and four launches with them.  Every output element is computed
    exactly as by `conv2d(x, p)`: bit-identical (tests/test_gpu_parity.py), so the choice may depend on the batch.  Falls back to one
    launch per map where the form does not apply (another arithmetic, a pinned / audited layer, channel counts the 128-tile blocks do
    not fit, more than five maps).  Inside `share_wino_input` every map's slice is published for the layer's co-readers."""
    xs = list(xs)
    ok = (WINO_LEVELS and DEFAULT_PRECISION == 3 and AUDIT is None and 2 <= len(xs) <= 5 and p.w_wino is not None
          and p.pin_precision != 2 and p.KH == 3 and p.stride == 1 and p.pad == 1 and not (p.stem or p.phase or p.pixshuf)
          and all(x.is_cuda and x.dtype == torch.float32 and x.shape[0] == xs[0].shape[0] and x.shape[3] == p.Cin for x in xs)
          and p.Cin % 32 == 0 and p.Cin >= 256 and -(-p.cols // 64) % 2 == 0)
    if ok and _WINO_SHARE is not None:  # (a co-reader already transformed one of the maps into a buffer of its own: keep the per-map form)
        ok = all((_WINO_SHARE.get(x.data_ptr()) or [None, None])[1] is None for x in xs)
    tiles = [x.shape[0] * ((x.shape[1] + 1) // 2) * ((x.shape[2] + 1) // 2) for x in xs]
    total = sum(tiles)
    if not ok or total * p.Cin * 4 > _ADDR_LIMIT:
        return [conv2d(x, p) for x in xs]
    dev = xs[0].device
    if getattr(p, "_u_scale", None) is None:
        p._u_scale = _pow2_scale_host(float(p.w_wino.abs().max()))
    if p.w_wino_h2 is None or p.w_wino_h2.device != p.w_wino.device:
        rows, cols = p.w_wino.shape[1], p.w_wino.shape[2]
        p.w_wino_h2 = torch.empty((16, cols // 32, 2, rows, 32), device=p.w_wino.device, dtype=torch.float16)
        _lib.check(_lib.lib().a3d_split_f16x2_chunk(p.w_wino.data_ptr(), p.w_wino_h2.data_ptr(), 16, rows, cols, 32, p._u_scale, _stream()), "a3d_split_f16x2_chunk")
        if not os.environ.get("A3D_NO_PUBLISH"):
            torch.cuda.current_stream().synchronize()
    ws = torch.empty(16 * total * p.Cin, device=dev, dtype=torch.float32)
    descs = (_lib.ConvDesc * len(xs))()
    outs, off = [], 0
    global _LAST_PRECISION
    _LAST_PRECISION = 3
    # measurement (bench.py): event pairs around the transforms and around the one GEMM launch, under the same rule as _conv2d_launch
    label = "wino_gemm_h2w_kernel<4>"  # (the same kernel as a single map's launch -- the table in its epilogue has five rows instead of one)
    timing = CONV_TIMING is not None and (CONV_TIMING_ONLY is None or label in CONV_TIMING_ONLY or "wino_input_kernel" in CONV_TIMING_ONLY)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if timing else None
    if timing:
        ev[0].record()
    for k, x in enumerate(xs):
        _req(x)
        B, H, W, Cin = x.shape
        out = torch.empty((B, H, W, p.cols), device=dev, dtype=torch.float32)
        d = descs[k]
        d.x, d.w, d.scale, d.shift, d.y = _p(x), _p(p.w), _p(p.scale), _p(p.shift), _p(out)
        d.B, d.H, d.W, d.Cin, d.Cin2 = B, H, W, Cin, 0
        d.Ho, d.Wo, d.Cout = H, W, p.cols
        d.KH, d.KW, d.stride, d.pad = 3, 3, 1, 1
        d.Kpad, d.ups, d.act = p.Kpad, 0, p.act
        d.res_ups, d.pixshuf, d.stem, d.splitk = 0, 0, 0, 1
        d.tune, d.phase, d.precision = 0, 0, 3
        d.w_wino, d.w_wino_x3, d.w_scale = p.w_wino.data_ptr(), p.w_wino_h2.data_ptr(), p._u_scale
        d.in_amax = amax_of(x).data_ptr()
        if not os.environ.get("A3D_NO_YAMAX"):
            out._a3d_amax = amax_slot(B, dev)
            d.y_amax = out._a3d_amax.data_ptr()
        d.workspace = ws.data_ptr()
        d.wino_t_off, d.wino_t_total = off, total
        _lib.check(_lib.lib().a3d_wino_input_transform(C.byref(d), _stream()), "a3d_wino_input_transform")
        if _WINO_SHARE is not None:
            ent = _WINO_SHARE.get(x.data_ptr())
            if ent is not None and ent[0].shape == x.shape:
                ent[1:] = [ws, 3, (off, total)]  # [tensor, V buffer, arithmetic of its format, (slice offset, tiles per run)]
        off += tiles[k]
        outs.append(out)
    if timing:
        ev[1].record()
    _lib.check(_lib.lib().a3d_wino_gemm_levels(descs, len(xs), _stream()), "a3d_wino_gemm_levels")
    if timing:
        ev[2].record()
        px = sum(x.shape[0] * x.shape[1] * x.shape[2] for x in xs)
        shape = f"{xs[0].shape[0]}x[{'+'.join(str(x.shape[1]) + 'x' + str(x.shape[2]) for x in xs)}]x{p.Cin}->{p.cols} k3 s1"
        CONV_TIMING.append(("wino_input_kernel", 0.0, ev[0], ev[1], shape, 0.0, "none", _stream()))
        CONV_TIMING.append((label, 2.0 * px * p.cols * 9 * p.Cin, ev[1], ev[2], shape, 2.0 * total * 16 * p.cols * p.Cin, "f16x3", _stream()))
    return outs


def conv2d(x: torch.Tensor, p: PackedConv, *, x2: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
           res_ups: bool = False, ups: bool = False, act: Optional[int] = None, splitk: int = 1,
           m_dev: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, tune: int = 0,
           wino: Optional[bool] = None, gate: Optional[torch.Tensor] = None, precision=None, out_dtype=None, dot=None) -> torch.Tensor:
    """x: NHWC [B,H,W,Cin] (stem: [B,H,W,4]).  Returns NHWC [B,Ho,Wo,cols] (pixshuf: [B,2Ho,2Wo,cols/4]).

    bf16 STORAGE (training step, precision 1 only): x / res / gate may be torch.bfloat16 tensors and `out_dtype=torch.bfloat16`
    makes the layer store its output as bf16 (a3d_conv_desc.io_bf16) -- half the HBM bytes of every such tensor.

    A launch addresses each operand through a 32-bit buffer descriptor.  Batches whose largest tensor (input, output, residual,
    Winograd tiles, split-K partial sums) reaches 4 GiB -- ~86 frames x 1000 proposals at the box head's fc1, ~218 frames at a
    256-channel 120x160 layer -- run as consecutive launches over blocks of images.  Every image's result is a function of that
    image alone (per-image scales, fixed layer algorithm), so the blocks reproduce the single launch bit for bit."""
    if AUDIT is not None and not AUDIT.busy and precision is None and x.dtype == torch.float32 and out_dtype is None and not p.phase:
        return AUDIT.conv2d(x, p, x2=x2, res=res, res_ups=res_ups, ups=ups, act=act, splitk=splitk, m_dev=m_dev, out=out, tune=tune, wino=wino, gate=gate)
    if x.dtype == torch.float16:
        # (pre-split activations: ONE kernel form reads them -- nothing it cannot honour may be dropped silently)
        if splitk != 1 or m_dev is not None or gate is not None or tune or out_dtype is not None or ups or dot is not None or wino or precision not in (None, 3):
            raise RuntimeError("pre-split (float16) activations run the dual-DMA fp16x2 kernel only: splitk, m_dev, gate, tune, out_dtype, ups, dot, "
                               "wino and any precision other than 3 are not available with them")
        return _conv2d_presplit(x, p, x2=x2, res=res, res_ups=res_ups, act=act, out=out)
    if out_dtype is not None or x.dtype == torch.bfloat16 or (res is not None and res.dtype == torch.bfloat16) or (gate is not None and gate.dtype == torch.bfloat16):
        return _conv2d_bf16_storage(x, p, res=res, res_ups=res_ups, act=act, out=out, gate=gate, precision=precision, out_dtype=out_dtype, tune=tune)
    _req(x)
    B = x.shape[0]
    if B > 1:
        Hl_, Wl_ = (2 * x.shape[1], 2 * x.shape[2]) if ups else (x.shape[1], x.shape[2])
        Ho_ = x.shape[1] if p.phase else (Hl_ + 2 * p.pad - p.KH) // p.stride + 1
        Wo_ = x.shape[2] if p.phase else (Wl_ + 2 * p.pad - p.KW) // p.stride + 1
        per_in = x.shape[1] * x.shape[2] * max(x.shape[3], 0 if x2 is None else x2.shape[3]) * 4
        per_out = Ho_ * Wo_ * p.cols * 4 * (4 if 0 < p.phase < 5 else 1) * max(1, int(splitk))
        if dot is not None:  # the tap-product epilogue stores nine planes of the 2x map, not the 64-channel tensor
            per_out = 9 * 4 * Ho_ * Wo_ * 4
        per_v = ((Hl_ + 1) // 2) * ((Wl_ + 1) // 2) * (x.shape[3] + (0 if x2 is None else x2.shape[3])) * 4 if p.w_wino is not None else 0
        per = max(per_in, per_out, per_v, 1)
        if B * per > _ADDR_LIMIT:
            if dot is not None:
                raise RuntimeError("tap-product epilogue (dot): the batch must fit one launch")
            return _conv2d_blocks(x, p, max(1, _ADDR_LIMIT // per), (Ho_, Wo_), x2=x2, res=res, res_ups=res_ups, ups=ups, act=act, splitk=splitk,
                                  m_dev=m_dev, out=out, tune=tune, wino=wino, gate=gate, precision=precision)
    return _conv2d_launch(x, p, x2=x2, res=res, res_ups=res_ups, ups=ups, act=act, splitk=splitk, m_dev=m_dev, out=out, tune=tune,
                          wino=wino, gate=gate, precision=precision, dot=dot)


def _conv2d_blocks(x, p, nb, hw_out, *, x2, res, out, gate, m_dev, **kw):
    B = x.shape[0]
    Ho, Wo = hw_out
    if out is None:
        shape = (B, 2 * Ho, 2 * Wo, p.cols // 4) if p.pixshuf else (B, Ho, Wo, p.cols)
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    cut = lambda t, s, e: None if t is None else keep_amax(t[s:e], _amax_rows(t, s, e))
    recorded = []
    for s in range(0, B, nb):
        e = min(s + nb, B)
        md = None if m_dev is None else (m_dev.reshape(1) - s).clamp(0, e - s).to(torch.int32)  # live rows of this block (device side)
        o = out[s:e]
        ya = getattr(out, "_a3d_amax", None)
        if ya is not None and ya.numel() == B:  # (the phase launches of an upsampled conv share their output's slots)
            o._a3d_amax = ya[s:e]
        _conv2d_launch(cut(x, s, e), p, x2=cut(x2, s, e), res=cut(res, s, e), out=o, gate=cut(gate, s, e), m_dev=md, **kw)
        recorded.append(getattr(o, "_a3d_amax", None))
    if getattr(out, "_a3d_amax", None) is None and all(r is not None for r in recorded):
        out._a3d_amax = torch.cat(recorded)
    return out


BF16_SPLITK = os.environ.get("A3D_BF16_SPLITK", "1") != "0"
# Auto split-K sizes the number of reduction splits by M = B * Ho * Wo: the bf16 summation order then depends on the batch.  That is the
# training step's business (DetectorTrainer.forward_backward turns it on for its launches; a training batch is one unit anyway) -- the
# opt-in bf16 INFERENCE mode (A3D_PRECISION=1 / bench.py --precision bf16) keeps one launch per layer, so that a frame's result stays
# independent of its batch in every mode and `conv2d`'s blocks-of-images form reproduces the single launch there too.
BF16_SPLITK_AUTO = False
# (tiles at most, workgroups aimed at, chunks per split at least, chunks at least, splits at most)
_BF16_SK_CFG = tuple(int(v) for v in os.environ.get("A3D_BF16_SPLITK_CFG", "128,256,12,24,8").split(","))


def _bf16_splitk(M: int, cols: int, Kpad: int) -> int:
    """Splits of the reduction for a bf16-arithmetic launch (csrc/conv_bf16.hip): layers whose 128 x 64 tiles leave most of the chip idle
    while each walks a long reduction one memory round trip at a time (the training step at the reference's 2 images per GPU: 3x3
    256 -> 256 on 2 x 30 x 40 pixels = 76 workgroups x 72 chunks).  A function of the layer's shape and the batch."""
    tiles = -(-M // 128) * -(-cols // 64)
    nk = Kpad // 32
    maxtiles, target, per, minnk, cap = _BF16_SK_CFG
    if not BF16_SPLITK or tiles > maxtiles or nk < minnk:
        return 1
    return int(max(1, min(cap, nk // per, target // tiles)))


_BF16_DESC_CACHE: dict = {}


def _conv2d_bf16_storage(x, p: PackedConv, *, res, res_ups, act, out, gate, precision, out_dtype, tune=0) -> torch.Tensor:
    """The training step's bf16 (autocast) arithmetic with tensors stored as bf16 where the caller says so: plain conv / linear
    layers only (no stem, upsampling, concat, split-K, Winograd), precision 1."""
    if precision != 1:
        raise RuntimeError("bf16-stored tensors belong to the bf16 arithmetic (precision=1): the fp32-grade modes keep fp32 tensors")
    ok = lambda t: t is None or (t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16))
    if not (ok(x) and ok(res) and ok(gate) and ok(out)):
        raise RuntimeError("expected contiguous fp32 / bf16 CUDA tensors")
    B, H, W, Cin = x.shape
    assert Cin == p.Cin and not (p.stem or p.pixshuf or p.phase), "plain layers only"
    Ho = (H + 2 * p.pad - p.KH) // p.stride + 1
    Wo = (W + 2 * p.pad - p.KW) // p.stride + 1
    if out is None:
        out = torch.empty((B, Ho, Wo, p.cols), device=x.device, dtype=out_dtype or torch.float32)
    if B * max(H * W * Cin, Ho * Wo * p.cols) * 4 > _ADDR_LIMIT:
        raise RuntimeError("tensor past the 32-bit addressing limit of one launch: split the batch")
    if gate is not None:
        assert tuple(gate.shape) == tuple(out.shape), (gate.shape, out.shape)
    # The descriptor's ~40 fields are the same from step to step for a given layer and batch shape: a filled prototype is kept per
    # (filter, shapes, storage types, flags) and only the tensor pointers are written per launch -- ~15 -> ~5 us of host time for each of the
    # ~120 such launches of a step, which matters at the reference's 2 images per GPU (5.4 ms per step, 4 of them host enqueue).
    act_eff = p.act if act is None else act
    key = (p.w.data_ptr(), _p(p.scale), _p(p.shift), _p(p.w_b16), p.KH, p.KW, p.stride, p.pad, p.Kpad, p.cols, act_eff, B, H, W, Cin,
           x.dtype, out.dtype, None if res is None else res.dtype, None if gate is None else gate.dtype, bool(res_ups), int(tune), BF16_SPLITK_AUTO, BF16_SPLITK)
    proto = _BF16_DESC_CACHE.get(key)
    if proto is None:
        d = _lib.ConvDesc()
        d.w, d.scale, d.shift = _p(p.w), _p(p.scale), _p(p.shift)
        d.B, d.H, d.W, d.Cin, d.Cin2 = B, H, W, Cin, 0
        d.Ho, d.Wo, d.Cout = Ho, Wo, p.cols
        d.KH, d.KW, d.stride, d.pad = p.KH, p.KW, p.stride, p.pad
        d.Kpad, d.ups, d.act = p.Kpad, 0, act_eff
        d.res_ups, d.pixshuf, d.stem, d.splitk = int(res_ups), 0, 0, 1
        d.precision = 1
        b16 = lambda t: t is not None and t.dtype == torch.bfloat16
        d.io_bf16 = (1 if b16(x) else 0) | (2 if b16(out) else 0) | (4 if b16(res) else 0) | (8 if b16(gate) else 0)
        d.tune = int(tune)
        if p.w_b16 is not None:  # the filter as bf16: large launches take both operands by LDS-DMA (csrc/conv_bf16w.hip; the same bits)
            d.w_bf16 = p.w_b16.data_ptr()
        sk = 1 if (res_ups or not BF16_SPLITK_AUTO) else _bf16_splitk(B * Ho * Wo, p.cols, p.Kpad)  # (split-K by batch size: inside the training step only)
        d.splitk = sk
        if len(_BF16_DESC_CACHE) > 4096:
            _BF16_DESC_CACHE.clear()
        proto = _BF16_DESC_CACHE[key] = (bytes(d), sk)
    d = _lib.ConvDesc.from_buffer_copy(proto[0])
    sk = proto[1]
    d.x, d.y = x.data_ptr(), out.data_ptr()
    if res is not None:
        d.res = res.data_ptr()
    if gate is not None:
        d.gate = gate.data_ptr()
    if sk > 1:
        ws = torch.empty(sk * B * Ho * Wo * p.cols, device=x.device, dtype=torch.float32)
        d.workspace = ws.data_ptr()
    if CONV_TIMING is not None:  # (tools/train_bench.py's roofline leg: these launches carry most of the bf16 step's FLOPs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(_lib.lib().a3d_conv2d_nhwc_f32(C.byref(d), _stream()), "a3d_conv2d_nhwc_f32")
        e1.record()
        fl = 2.0 * B * Ho * Wo * p.cols * p.KH * p.KW * p.Cin
        CONV_TIMING.append((last_conv_variant(), fl, e0, e1, f"{B}x{H}x{W}x{Cin}->{p.cols} k{p.KH} s{p.stride} io{int(d.io_bf16)}", fl, "bf16", _stream()))
        return out
    _lib.check(_lib.lib().a3d_conv2d_nhwc_f32(C.byref(d), _stream()), "a3d_conv2d_nhwc_f32")
    return out


def presplit_f16x2(x: torch.Tensor, x2: Optional[torch.Tensor] = None):
    """x [B, H, W, C] fp32 (C % 16 == 0) -> torch.float16 [B, H, W, C/16, 2, 16]: the two fp16 planes of x * s(b), s(b) from the tensor's
    recorded per-image maxima -- a3d_conv_desc.x_h2 (a3d_presplit_f16x2).  With x2 (second source of a channel concat) both tensors are
    split under the scale of max(amax(x)[b], amax(x2)[b]), as the consuming kernel scales them; returns the pair."""
    _req(x)
    B, n = x.shape[0], x.numel() // x.shape[0]
    ax, ax2 = amax_of(x), (None if x2 is None else amax_of(_req(x2)))
    outs = []
    for t in ((x,) if x2 is None else (x, x2)):
        nt = t.numel() // B
        o = torch.empty(tuple(t.shape[:-1]) + (t.shape[-1] // 16, 2, 16), device=t.device, dtype=torch.float16)
        _lib.check(_lib.lib().a3d_presplit_f16x2(t.data_ptr(), o.data_ptr(), ax.data_ptr(), _p(ax2), B, nt, _stream()), "a3d_presplit_f16x2")
        o._a3d_amax = ax if x2 is None else torch.maximum(ax, ax2)
        outs.append(o)
    return outs[0] if x2 is None else tuple(outs)


def _conv2d_presplit(x, p: PackedConv, *, x2=None, res=None, res_ups=False, act=None, out=None) -> torch.Tensor:
    """A direct fp16x2 layer on PRE-SPLIT activations (x, x2: torch.float16 [B, H, W, C/16, 2, 16] with recorded maxima): the wide kernel
    with both operands by LDS-DMA ("conv_h2w_kernel xd").  Bit-identical to conv2d on the fp32 tensor the planes were split from."""
    if DEFAULT_PRECISION != 3 and not os.environ.get("A3D_ALLOW_PRESPLIT"):
        raise RuntimeError("pre-split activations belong to the fp16x2 arithmetic (A3D_PRECISION=3)")
    ok = lambda t: t.is_cuda and t.is_contiguous() and t.dtype == torch.float16 and t.dim() == 6 and t.shape[4:] == (2, 16)
    if not ok(x) or (x2 is not None and not ok(x2)):
        raise RuntimeError("expected contiguous pre-split fp16 tensors [B, H, W, C/16, 2, 16]")
    B, H, W = x.shape[:3]
    Cin = x.shape[3] * 16
    Cin2 = 0 if x2 is None else x2.shape[3] * 16
    assert Cin + Cin2 == p.Cin and not (p.stem or p.pixshuf or p.phase) and p.presplit and p.Kpad == p.KH * p.KW * p.Cin, "plain direct layers only"
    Ho = (H + 2 * p.pad - p.KH) // p.stride + 1
    Wo = (W + 2 * p.pad - p.KW) // p.stride + 1
    if B * max(H * W * max(Cin, Cin2), Ho * Wo * p.cols) * 4 > _ADDR_LIMIT:
        # (blocks of images, as conv2d does for fp32 tensors: every image's result is a function of that image alone)
        nb = max(1, _ADDR_LIMIT // (max(H * W * max(Cin, Cin2), Ho * Wo * p.cols) * 4))
        if out is None:
            out = torch.empty((B, Ho, Wo, p.cols), device=x.device, dtype=torch.float32)
        ya = amax_slot(B, x.device)
        cut = lambda t, s0, e0: None if t is None else keep_amax(t[s0:e0], _amax_rows(t, s0, e0))
        for s0 in range(0, B, nb):
            e0 = min(s0 + nb, B)
            o = out[s0:e0]
            o._a3d_amax = ya[s0:e0]
            _conv2d_presplit(cut(x, s0, e0), p, x2=cut(x2, s0, e0), res=cut(res, s0, e0), res_ups=res_ups, act=act, out=o)
        out._a3d_amax = ya
        return out
    if out is None:
        out = torch.empty((B, Ho, Wo, p.cols), device=x.device, dtype=torch.float32)
    d = _lib.ConvDesc()
    d.x_h2, d.x2_h2 = x.data_ptr(), _p(x2)
    d.w, d.scale, d.shift, d.res, d.y = _p(p.w), _p(p.scale), _p(p.shift), _p(res), _p(out)
    d.B, d.H, d.W, d.Cin, d.Cin2 = B, H, W, Cin, Cin2
    d.Ho, d.Wo, d.Cout = Ho, Wo, p.cols
    d.KH, d.KW, d.stride, d.pad = p.KH, p.KW, p.stride, p.pad
    d.Kpad, d.ups, d.act = p.Kpad, 0, p.act if act is None else act
    d.res_ups, d.pixshuf, d.stem, d.splitk = int(res_ups), 0, 0, 1
    d.precision = 3
    d.tune = int(os.environ.get("A3D_XD_TUNE", "0"))  # (developer builds with -DA3D_ABLATIONS only: timing-only variants of the ring)
    a = getattr(x, "_a3d_amax", None)
    if a is None or a.numel() != B:
        raise RuntimeError("a pre-split tensor carries the per-image maxima it was scaled by (_a3d_amax)")
    d.in_amax = a.data_ptr()  # (two sources: both were split under the shared maximum, recorded on each)
    if getattr(p, "_w_scale", None) is None:
        p._w_scale = _pow2_scale_host(float(p.w.abs().max()))
    d.w_scale = p._w_scale
    if p.w_h2 is None or p.w_h2.device != p.w.device:
        p.w_h2 = torch.empty((p.Kpad // 16, 2, p.w.shape[0], 16), device=p.w.device, dtype=torch.float16)
        _lib.check(_lib.lib().a3d_split_f16x2_chunk(p.w.data_ptr(), p.w_h2.data_ptr(), 1, p.w.shape[0], p.Kpad, 16, d.w_scale, _stream()), "a3d_split_f16x2_chunk")
        if not os.environ.get("A3D_NO_PUBLISH"):
            torch.cuda.current_stream().synchronize()
    d.w_x3 = p.w_h2.data_ptr()
    if not os.environ.get("A3D_NO_YAMAX"):
        ya = getattr(out, "_a3d_amax", None)
        if ya is None or ya.numel() != out.shape[0]:
            ya = out._a3d_amax = amax_slot(out.shape[0], out.device)
        d.y_amax = ya.data_ptr()
    timing = CONV_TIMING is not None
    if timing and CONV_TIMING_ONLY is not None:
        timing = "conv_h2w_kernel xd" in CONV_TIMING_ONLY
    if timing:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.lib().a3d_conv2d_nhwc_f32(C.byref(d), _stream()), "a3d_conv2d_nhwc_f32")
    if timing:
        e1.record()
        fl = 2.0 * B * Ho * Wo * p.cols * p.KH * p.KW * p.Cin
        CONV_TIMING.append((last_conv_variant(), fl, e0, e1, f"{B}x{H}x{W}x{Cin + Cin2}->{p.cols} k{p.KH} s{p.stride} presplit", fl, "f16x3", _stream()))
    return out


class _amax_rows:
    """Stand-in `old` argument of keep_amax for a block of images: rows [s, e) of a tensor's recorded maxima."""

    def __init__(self, t, s, e):
        a = getattr(t, "_a3d_amax", None)
        self._a3d_amax = None if a is None or a.numel() != t.shape[0] else a[s:e]
        self.shape = (e - s,)


def _conv2d_launch(x: torch.Tensor, p: PackedConv, *, x2: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
                   res_ups: bool = False, ups: bool = False, act: Optional[int] = None, splitk: int = 1,
                   m_dev: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, tune: int = 0,
                   wino: Optional[bool] = None, gate: Optional[torch.Tensor] = None, precision=None, dot=None) -> torch.Tensor:
    _req(x)
    # Launch plans.  For a module-cached layer called the way the detector calls it (no options, output allocated here) everything in the
    # descriptor but six tensor pointers is a function of (layer, input shape, residual form, arithmetic mode): the finished descriptor is
    # kept on the PackedConv and later calls write the pointers into a copy -- ~15 -> ~6 us of host time per launch.  The reference's
    # per-frame loop makes ~100 such calls per frame and its single-frame pass is as long on the host as on the GPU (MEASUREMENTS.md,
    # round 6, item 12).  Anything else takes the full path below, which is also what creates the plan.
    plan_key = None
    if (LAUNCH_PLANS and CONV_TIMING is None and x2 is None and m_dev is None and out is None and gate is None and dot is None and not ups and splitk == 1
            and tune == 0 and wino is None and precision is None and act is None and not (p.phase or p.pixshuf or p.stem) and p.presplit
            and (_WINO_SHARE is None or x.data_ptr() not in _WINO_SHARE)):
        plan_key = (x.shape, res is not None, bool(res_ups), DEFAULT_PRECISION, p.pin_precision, WINO_PLANE_SPLIT, WINO_TUNE, WINO_MAX_HW, BF16_SPLITK_AUTO,
                    x.device, p.w.data_ptr())
        plans = p.__dict__.get("_plans")
        plan = plans.get(plan_key) if plans else None
        if plan is not None:
            proto, oshape, need_in, need_y, ws_n, m_n, prec = plan
            out = torch.empty(oshape, device=x.device, dtype=torch.float32)
            d = _lib.ConvDesc.from_buffer_copy(proto)
            d.x, d.y = x.data_ptr(), out.data_ptr()
            if res is not None:
                d.res = _req(res).data_ptr()
            if need_in:
                d.in_amax = amax_of(x).data_ptr()
            if need_y:
                ya = out._a3d_amax = amax_slot(oshape[0], x.device)
                d.y_amax = ya.data_ptr()
            if ws_n:
                ws = torch.empty(ws_n, device=x.device, dtype=torch.float32)
                d.workspace = ws.data_ptr()
            if m_n:
                wino_m = torch.empty(m_n, device=x.device, dtype=torch.float32)
                d.wino_m = wino_m.data_ptr()
            global _LAST_PRECISION
            _LAST_PRECISION = prec
            _lib.check(_lib.lib().a3d_conv2d_nhwc_f32(C.byref(d), _stream()), "a3d_conv2d_nhwc_f32")
            return out
    B, H, W, Cin = x.shape
    Cin2 = 0
    if x2 is not None:
        _req(x2)
        Cin2 = x2.shape[3]
        assert x2.shape[:3] == x.shape[:3]
    if not p.stem:
        assert Cin + Cin2 == p.Cin, (Cin, Cin2, p.Cin)
    Hl, Wl = (2 * H, 2 * W) if ups else (H, W)
    Ho = (Hl + 2 * p.pad - p.KH) // p.stride + 1
    Wo = (Wl + 2 * p.pad - p.KW) // p.stride + 1
    if p.phase:
        assert not ups and out is not None, "phase convs run on the source grid and fill a shared [B,2H,2W,C] output"
        Ho, Wo = H, W
        if p.phase == 5:
            assert precision in (2, 3) and (dot is not None or tuple(out.shape) == (B, 2 * H, 2 * W, p.cols // 4)), "the fused four-phase form belongs to the split-operand arithmetics"
    if out is None:
        shape = (B, 2 * Ho, 2 * Wo, p.cols // 4) if p.pixshuf else (B, Ho, Wo, p.cols)
        out = torch.empty(shape, device=x.device, dtype=torch.float32)
    d = _lib.ConvDesc()
    d.x, d.x2, d.w, d.scale, d.shift, d.res, d.y = _p(x), _p(x2), _p(p.w), _p(p.scale), _p(p.shift), _p(res), _p(out)
    d.B, d.H, d.W, d.Cin, d.Cin2 = B, H, W, Cin, Cin2
    d.Ho, d.Wo, d.Cout = Ho, Wo, p.cols
    d.KH, d.KW, d.stride, d.pad = p.KH, p.KW, p.stride, p.pad
    d.Kpad, d.ups, d.act = p.Kpad, int(ups), p.act if act is None else act
    d.res_ups, d.pixshuf, d.stem, d.splitk = int(res_ups), int(p.pixshuf), int(p.stem), int(splitk)
    d.m_dev = _p(m_dev)
    if gate is not None:
        assert tuple(_req(gate).shape) == tuple(out.shape), (gate.shape, out.shape)
        d.gate = gate.data_ptr()
    d.tune = int(tune)
    if dot is not None:  # (phase 5 only: the nine tap products of a following 3x3 convolution to one channel instead of the output)
        assert p.phase == 5 and precision == 3
        d.dot_w, d.dot_y = _req(dot[0]).data_ptr(), _req(dot[1]).data_ptr()
        d.y = None  # (`out` is the tap-product tensor itself, kept as the handle of the recorded maxima: the [B,2H,2W,C] output is never formed)
    wino_ok = (p.w_wino is not None and res is None and splitk == 1 and m_dev is None and (tune in (0, 7, 8, 23, 24, 25) or tune >= 200) and not ups
               and (wino if wino is not None else True))
    if precision is None or precision == "bf16x3":  # a MODE (module default, or the caller's "bf16x3"): pick per layer kind
        mode = DEFAULT_PRECISION if precision is None else 2
        if mode == 3 and p.pin_precision == 2:  # pinned by the precision audit (a function of the LAYER: batch invariance survives)
            mode = 2
        plain = not (p.stem or ups or p.phase or p.pixshuf or x2 is not None or splitk != 1 or m_dev is not None) and p.Kpad == p.KH * p.KW * p.Cin
        if mode == 1:
            precision = 1 if plain and p.Cin % 32 == 0 else 0
        elif mode in (2, 3):  # a function of the layer only, like the Winograd rule (batch-size invariance)
            # (the bf16x3 kernel also takes the phase convs of the depth decoder and their 2-source channel concat)
            # (split-K -- the 50176-deep head FCs -- only through the wide kernel, whose conditions the last line repeats)
            x3_ok = p.stem and not ups and splitk == 1 and m_dev is None and x2 is None and res is None
            x3_ok = x3_ok or (not (p.stem or ups or m_dev is not None) and p.Kpad == p.KH * p.KW * p.Cin
                     and not (p.pixshuf and (res is not None or gate is not None))
                     and (splitk == 1 or (p.presplit and p.Kpad >= 4096 and p.cols >= 192 and -(-p.cols // 256) * 256 <= p.cols + p.cols // 4
                                          and not p.phase and gate is None and not p.pixshuf))
                     and (x2 is None or Cin2 == Cin) and Cin % 16 == 0 and not (p.phase and res is not None))
            # Winograd layers keep the Winograd form with the split-operand GEMM (conv_wino.hip 2x, 32-deep chunks).
            # (Measured and NOT taken: the 64-channel 3x3 layers of res2 are 0.18 ms faster each in the one-launch fp32 Winograd
            # kernel -- 986 vs 971 frames/s -- but with that mix one of the 800 scores of the end-to-end test at threshold 0.0 moved
            # to 1.007e-4 from the oracle's, past the stated 1e-4: the arithmetic stays uniform.)
            precision = 2 if tune == 0 and ((x3_ok and not wino_ok) or (wino_ok and (Cin + Cin2) % 32 == 0)) else 0
            if precision == 2 and mode == 3 and H3_KINDS(p, wino_ok, splitk):
                precision = 3
        else:
            precision = 0
    d.precision = int(precision)
    if d.precision == 1 and p.w_b16 is not None:
        d.w_bf16 = p.w_b16.data_ptr()
    if (d.precision == 1 and splitk == 1 and BF16_SPLITK_AUTO and not (p.stem or ups or p.phase or p.pixshuf or x2 is not None or m_dev is not None or res_ups)
            and p.Kpad == p.KH * p.KW * p.Cin and p.Cin % 32 == 0):
        splitk = _bf16_splitk(B * Ho * Wo, p.cols, p.Kpad)  # (bf16 arithmetic: small grids with long reductions)
        d.splitk = splitk
    _LAST_PRECISION = int(precision)
    d.phase = int(p.phase)
    if d.precision == 3:  # fp16x2: per-image scales of the activations (recorded by their producers), one static scale of the filter
        d.in_amax = amax_of(x).data_ptr()
        if x2 is not None:
            d.in_amax2 = amax_of(x2).data_ptr()
        if getattr(p, "_w_scale", None) is None:
            p._w_scale = _pow2_scale_host(float(p.w.abs().max()))
        d.w_scale = p._w_scale
    if d.precision in (2, 3) and (DEFAULT_PRECISION == 3 or d.precision == 3) and not p.pixshuf and not os.environ.get("A3D_NO_YAMAX"):
        ya = getattr(out, "_a3d_amax", None)  # (the four phase launches of an upsampled conv share their output and its slot)
        if ya is None or ya.numel() != out.shape[0]:
            ya = out._a3d_amax = amax_slot(out.shape[0], out.device)
        d.y_amax = ya.data_ptr()
    # Winograd F(2x2,3x3) for every 3x3 s1 p1 layer that has Winograd-domain weights.  The choice must not depend on the
    # batch / ROI count (a frame's result would otherwise depend on how it was batched), so it is a function of the layer
    # only; measured faster than the direct form down to the 8x10 level (tools/conv_bench.py: res5 0.50 -> 0.30 ms,
    # p5 RPN conv 0.18 -> 0.10 ms, res2 64->64 0.40 -> 0.38 ms per 32 frames).  `wino=False` forces the direct form.
    use_wino = wino_ok and (precision == 0 or (precision in (2, 3) and tune in (0, 8, 23, 24, 25) and (Cin + Cin2) % 32 == 0))
    if use_wino and precision == 3 and (-(-p.cols // 64) % 2 or (Cin + Cin2 < 256 and wino is None)):
        # fp16x2: the direct form where the wide Winograd kernels' 128-channel tiles do not fit (res2's 64 -> 64) and for layers under
        # 256 input channels -- with three MFMAs per k step the 16-plane round trip costs more than the 2.25x multiply-adds it saves
        # (whole-layer times at 64 frames, tools/h2_wino_vs_direct.py: 60x80x128 -> 128 0.435 Winograd | 0.352 direct; p2 256 -> 256
        # 3.99 | 4.85).  A function of the layer only.
        use_wino = False
    if use_wino and precision == 3 and WINO_MAX_HW and H * W > WINO_MAX_HW and p.presplit and (-(-p.cols // 256) * 256 <= p.cols + p.cols // 4):
        # (experiment knob, a function of the layer and the image size only: 3x3 layers on maps larger than A3D_WINO_MAX_HW pixels take
        # the wide DIRECT kernel -- same time as Winograd alone on the p2 / p3 levels, a fifth of its HBM traffic)
        use_wino = False
        d.tune = 9
    ws = None
    if use_wino:
        if WINO_TUNE and d.tune == 0 and d.precision == 3:
            d.tune = WINO_TUNE
        d.w_wino = p.w_wino.data_ptr()
        if p.w_wino_cm is not None and d.precision == 0 and tune == 0:
            d.w_wino_cm = p.w_wino_cm.data_ptr()  # the library then takes the one-launch kernel where the layer qualifies
        if d.precision == 3:  # the Winograd-domain filter as two fp16 planes, scaled by the power of two of ITS maximum
            if getattr(p, "_u_scale", None) is None:
                p._u_scale = _pow2_scale_host(float(p.w_wino.abs().max()))
            d.w_scale = p._u_scale
            if p.w_wino_h2 is None or p.w_wino_h2.device != p.w_wino.device:
                rows, cols = p.w_wino.shape[1], p.w_wino.shape[2]
                p.w_wino_h2 = torch.empty((16, cols // 32, 2, rows, 32), device=p.w_wino.device, dtype=torch.float16)
                _lib.check(_lib.lib().a3d_split_f16x2_chunk(p.w_wino.data_ptr(), p.w_wino_h2.data_ptr(), 16, rows, cols, 32, p._u_scale, _stream()),
                           "a3d_split_f16x2_chunk")
                if not os.environ.get("A3D_NO_PUBLISH"):
                    torch.cuda.current_stream().synchronize()
            d.w_wino_x3 = p.w_wino_h2.data_ptr()
        if d.precision == 2:
            if p.w_wino_x3 is None or p.w_wino_x3.device != p.w_wino.device:  # once per layer
                rows, cols = p.w_wino.shape[1], p.w_wino.shape[2]
                p.w_wino_x3 = torch.empty((16, cols // 32, 3, rows, 32), device=p.w_wino.device, dtype=torch.bfloat16)
                _lib.check(_lib.lib().a3d_split_bf16x3(p.w_wino.data_ptr(), p.w_wino_x3.data_ptr(), 16, rows, cols, _stream()), "a3d_split_bf16x3")
                # the cache is read by later launches on ANY stream (the RPN head runs one packed layer on three streams for 1-2
                # frame batches): it must be complete before it is published.  Once per packed layer.
                if not os.environ.get("A3D_NO_PUBLISH"):
                    torch.cuda.current_stream().synchronize()
            d.w_wino_x3 = p.w_wino_x3.data_ptr()
    if d.precision == 3 and not use_wino and p.presplit and p.Kpad % 16 == 0:
        # every direct fp16x2 launch of a module-cached layer streams its filter pre-split by LDS-DMA (narrow and wide kernels alike)
        if p.w_h2 is None or p.w_h2.device != p.w.device:  # the fp16x2 planes of the filter, scaled by w_scale (once per packed layer)
            p.w_h2 = torch.empty((p.Kpad // 16, 2, p.w.shape[0], 16), device=p.w.device, dtype=torch.float16)
            _lib.check(_lib.lib().a3d_split_f16x2_chunk(p.w.data_ptr(), p.w_h2.data_ptr(), 1, p.w.shape[0], p.Kpad, 16, d.w_scale, _stream()),
                       "a3d_split_f16x2_chunk")
            if not os.environ.get("A3D_NO_PUBLISH"):
                torch.cuda.current_stream().synchronize()  # published to every stream, see w_wino_x3 below
        d.w_x3 = p.w_h2.data_ptr()
    if (d.precision == 2 and not use_wino and p.presplit and p.cols >= 192 and p.Kpad % 16 == 0 and (p.Kpad >= 4096 or tune == 9 or p.phase == 5)
            and not (p.stem or p.pixshuf)):
        # wide layers: weight planes pre-split once per packed layer, streamed by LDS-DMA (csrc/conv_bf16x3_wide.hip)
        if p.w_x3 is None or p.w_x3.device != p.w.device:
            p.w_x3 = torch.empty((p.Kpad // 16, 3, p.w.shape[0], 16), device=p.w.device, dtype=torch.bfloat16)
            _lib.check(_lib.lib().a3d_split_bf16x3_chunk(p.w.data_ptr(), p.w_x3.data_ptr(), 1, p.w.shape[0], p.Kpad, 16, _stream()), "a3d_split_bf16x3_chunk")
            if not os.environ.get("A3D_NO_PUBLISH"):
                torch.cuda.current_stream().synchronize()  # published to every stream, see w_wino_x3 above
        d.w_x3 = p.w_x3.data_ptr()
    fused_wino = False
    shared = None  # [input tensor, its transformed tiles V or None]: see share_wino_input
    nbytes = mbytes = 0
    if use_wino or splitk > 1:
        if use_wino and _WINO_SHARE is not None and x2 is None:
            shared = _WINO_SHARE.get(x.data_ptr())
            if shared is not None and (shared[0].shape != x.shape or d.tune not in (0, 23, 24)):  # (23 | 24: other loops of the same GEMM on the same V)
                shared = None
            # (the tiles' FORMAT belongs to the arithmetic: fp16x2 consumers read V pre-split into fp16 planes, the others fp32)
            if shared is not None and shared[1] is not None and len(shared) > 2 and shared[2] != int(d.precision):
                shared = None
        if shared is not None:
            d.w_wino_cm = None  # consumers of a shared input take the two-launch form so that V exists once for all of them
            if len(shared) > 3 and shared[1] is not None:  # (the tensor's tiles are a slice of a multi-level buffer: conv2d_levels)
                d.wino_t_off, d.wino_t_total = shared[3]
        nbytes = _lib.lib().a3d_conv_workspace_bytes(C.byref(d))
        fused_wino = use_wino and nbytes == 0  # the one-launch Winograd kernel needs no V tensor
        if shared is not None and shared[1] is not None:
            ws = shared[1]
            assert ws.numel() * 4 == nbytes
            d.workspace = ws.data_ptr()
        elif nbytes:
            ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
            d.workspace = ws.data_ptr()
        if use_wino and d.precision == 3 and WINO_PLANE_SPLIT:  # small problems: scratch for the plane-split form (a3d_conv_desc.wino_m; 0 bytes otherwise)
            mbytes = _lib.lib().a3d_wino_m_bytes(C.byref(d))
            if mbytes:
                wino_m = torch.empty(mbytes // 4, device=x.device, dtype=torch.float32)
                d.wino_m = wino_m.data_ptr()
    if shared is not None:
        have_v, shared[1] = shared[1] is not None, ws
        if len(shared) > 2:
            shared[2] = int(d.precision)
        else:
            shared.append(int(d.precision))
    # CONV_TIMING_ONLY (a set of variant labels): event pairs only around the launches whose label -- remembered from the last fully
    # instrumented pass over the same layer and shape -- is in the set; every other launch goes out as if nothing were measured
    # (700 event pairs per 64-frame step cost 0.8 ms of its 44).
    timing = CONV_TIMING is not None
    tkey = None
    if timing:
        tkey = (id(p), B, H, W, bool(ups), int(splitk), int(d.precision), int(d.tune), res is not None, x2 is not None, shared is not None)
        if CONV_TIMING_ONLY is not None:
            known = _TIMED_VARIANTS.get(tkey)
            timing = known is None or bool(known & CONV_TIMING_ONLY)
    if shared is not None and not timing:
        if not have_v:
            _lib.check(_lib.lib().a3d_wino_input_transform(C.byref(d), _stream()), "a3d_wino_input_transform")
        _lib.check(_lib.lib().a3d_wino_gemm(C.byref(d), _stream()), "a3d_wino_gemm")
        return out
    if timing:
        n_before = len(CONV_TIMING)
        try:
            return _conv2d_timed(d, p, out, shared, have_v if shared is not None else False, use_wino, fused_wino, B, H, W, Ho, Wo, Cin, Cin2, ups, splitk)
        finally:
            _TIMED_VARIANTS[tkey] = frozenset(t[0] for t in CONV_TIMING[n_before:])
    _lib.check(_lib.lib().a3d_conv2d_nhwc_f32(C.byref(d), _stream()), "a3d_conv2d_nhwc_f32")
    if plan_key is not None and shared is None:
        if p.__dict__.get("_plans") is None or len(p._plans) > 64:
            p._plans = {}
        p._plans[plan_key] = (bytes(d), tuple(out.shape), bool(d.in_amax), bool(d.y_amax), nbytes // 4 if ws is not None else 0,
                              mbytes // 4 if d.wino_m else 0, int(precision))
    return out


LAUNCH_PLANS = os.environ.get("A3D_LAUNCH_PLANS", "1") != "0"  # (False: every launch fills its descriptor from scratch; the same launches)
B2B_FUSED = os.environ.get("A3D_B2B", "1") != "0"  # (False: conv3 and the next block's conv1 as two launches)
B2B_MIN_PIXELS = 128 * 512  # one round of the chip's 512 workgroup slots (the activation-stationary kernel's own rule)
# (input channels of the first layer, output channels of the second) of the pairs that run as one launch: res2's conv3 -> conv1 boundaries
# (64 frames: 0.714 ms against 0.561 + 0.341 for the two launches).  The library also has the res3 form (128, 128): bit-identical as well,
# but its second GEMM (512 -> 128, six products per multiply-add) leaves 0.484 ms against 0.303 + 0.203 -- a tie, not taken.
B2B_PAIRS = ((64, 64),)


def conv2d_b2b(x: torch.Tensor, p1: PackedConv, res: torch.Tensor, p2: PackedConv):
    """conv3 (+ FrozenBN + residual + ReLU) of a bottleneck block and conv1 (+ FrozenBN + ReLU) of the NEXT block in one launch
    (csrc/conv_xs_b2b.hip, include/a3d.h a3d_conv_b2b): returns (y, z) = (conv2d(x, p1, res=res), conv2d(y, p2)) or None when the pair is
    not of that form (the caller then issues the two launches).  y and its maxima are bit for bit the single launch's; z is the second
    layer in the bf16x3 arithmetic (full 24-bit operands: the fp16x2 split of y would need y's per-image maximum, which no workgroup
    knows before the launch ends), bit for bit `conv2d(y, p2, precision=2)`.  Which form runs is a function of the layers and the map,
    never of the batch beyond the one-round bound every kernel choice of this size class has."""
    if not B2B_FUSED or DEFAULT_PRECISION != 3 or AUDIT is not None or not x.is_cuda or x.dtype != torch.float32:
        return None
    B, H, W, Cin = x.shape
    M = B * H * W
    if (p1.KH, p1.KW, p1.stride, p1.pad, p2.KH, p2.KW, p2.stride, p2.pad) != (1, 1, 1, 0, 1, 1, 1, 0) or p1.pin_precision == 2:
        return None
    if (Cin, p2.cols) not in B2B_PAIRS or p1.Cin != Cin or p2.Cin != p1.cols or p1.cols % 64 or not (p1.presplit and p2.presplit):
        return None
    if p1.Kpad != Cin or p2.Kpad != p2.Cin or p1.pixshuf or p2.pixshuf or p1.phase or p2.phase or p1.stem or p2.stem:
        return None
    if M < B2B_MIN_PIXELS or M * p1.cols * 4 >= (1 << 31) or res is None or tuple(res.shape) != (B, H, W, p1.cols):
        return None
    _req(x)
    _req(res)
    y = torch.empty((B, H, W, p1.cols), device=x.device, dtype=torch.float32)
    z = torch.empty((B, H, W, p2.cols), device=x.device, dtype=torch.float32)
    d1, d2 = _lib.ConvDesc(), _lib.ConvDesc()
    for d, p, xi, yo in ((d1, p1, x, y), (d2, p2, y, z)):
        d.x, d.w, d.scale, d.shift, d.y = _p(xi), _p(p.w), _p(p.scale), _p(p.shift), _p(yo)
        d.B, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = B, H, W, p.Cin, H, W, p.cols
        d.KH, d.KW, d.stride, d.pad, d.Kpad, d.act, d.splitk = 1, 1, 1, 0, p.Kpad, p.act, 1
        ya = yo._a3d_amax = amax_slot(B, x.device)
        d.y_amax = ya.data_ptr()
    d1.res = res.data_ptr()
    d1.precision, d2.precision = 3, 2
    d1.in_amax = amax_of(x).data_ptr()
    if getattr(p1, "_w_scale", None) is None:
        p1._w_scale = _pow2_scale_host(float(p1.w.abs().max()))
    d1.w_scale = p1._w_scale
    fresh = False
    if p1.w_h2 is None or p1.w_h2.device != p1.w.device:  # (as _conv2d_launch: the fp16x2 planes of the filter, once per packed layer)
        p1.w_h2 = torch.empty((p1.Kpad // 16, 2, p1.w.shape[0], 16), device=p1.w.device, dtype=torch.float16)
        _lib.check(_lib.lib().a3d_split_f16x2_chunk(p1.w.data_ptr(), p1.w_h2.data_ptr(), 1, p1.w.shape[0], p1.Kpad, 16, d1.w_scale, _stream()), "a3d_split_f16x2_chunk")
        fresh = True
    if p2.w_x3 is None or p2.w_x3.device != p2.w.device:  # the second filter's exact bf16 planes, once per packed layer
        p2.w_x3 = torch.empty((p2.Kpad // 16, 3, p2.w.shape[0], 16), device=p2.w.device, dtype=torch.bfloat16)
        _lib.check(_lib.lib().a3d_split_bf16x3_chunk(p2.w.data_ptr(), p2.w_x3.data_ptr(), 1, p2.w.shape[0], p2.Kpad, 16, _stream()), "a3d_split_bf16x3_chunk")
        fresh = True
    if fresh and not os.environ.get("A3D_NO_PUBLISH"):
        torch.cuda.current_stream().synchronize()  # (published to every stream, as the other per-layer caches)
    d1.w_x3, d2.w_x3 = p1.w_h2.data_ptr(), p2.w_x3.data_ptr()
    global _LAST_PRECISION
    _LAST_PRECISION = 3
    timing = CONV_TIMING is not None and (CONV_TIMING_ONLY is None or any(v.startswith("conv_h2xs_b2b") for v in CONV_TIMING_ONLY))
    if timing:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = _lib.lib().a3d_conv_b2b(C.byref(d1), C.byref(d2), _stream())
    if rc == -3:  # A3D_ERR_UNSUPPORTED: not a pair of this form after all
        return None
    _lib.check(rc, "a3d_conv_b2b")
    if timing:
        e1.record()
        fl = 2.0 * M * (p1.cols * Cin + p2.cols * p2.Cin)
        # (executed matrix FLOPs: 3 fp16 products per multiply-add of the first layer, 6 bf16 ones of the second -- booked at the fp16x2
        # weight of 3 with the second layer counted twice, so that the roofline object's issued-FLOP sum stays right)
        ex = 2.0 * M * (p1.cols * Cin + 2 * p2.cols * p2.Cin)
        CONV_TIMING.append((last_conv_variant(), fl, e0, e1, f"{B}x{H}x{W}x{Cin}->{p1.cols}->{p2.cols} k1 b2b", ex, "f16x3", _stream()))
    return y, z


CONV_TIMING_ONLY: Optional[frozenset] = None
_TIMED_VARIANTS: dict = {}


def _conv2d_timed(d, p, out, shared, have_v, use_wino, fused_wino, B, H, W, Ho, Wo, Cin, Cin2, ups, splitk):
    """The launch of _conv2d_launch with HIP events around each kernel (CONV_TIMING)."""
    k_real = 147 if p.stem else p.KH * p.KW * p.Cin
    shape = f"{B}x{H}x{W}x{Cin + Cin2}->{p.cols} k{p.KH} s{p.stride}{' ups' if ups else ''}{' sk%d' % splitk if splitk > 1 else ''}"
    ev = lambda: torch.cuda.Event(enable_timing=True)
    tiles = B * ((Ho + 1) // 2) * ((Wo + 1) // 2)
    if fused_wino:
        e0, e1 = ev(), ev()
        e0.record()
        _lib.check(_lib.lib().a3d_conv2d_nhwc_f32(C.byref(d), _stream()), "a3d_conv2d_nhwc_f32")
        e1.record()
        CONV_TIMING.append((last_conv_variant(), 2.0 * B * Ho * Wo * p.cols * k_real, e0, e1, shape, 2.0 * tiles * 16 * p.cols * p.Cin, "f32", _stream()))
        return out
    if use_wino:  # the two launches of the Winograd form are timed separately (they are separate kernels)
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        if shared is None or not have_v:
            _lib.check(_lib.lib().a3d_wino_input_transform(C.byref(d), _stream()), "a3d_wino_input_transform")
        e1.record()
        _lib.check(_lib.lib().a3d_wino_gemm(C.byref(d), _stream()), "a3d_wino_gemm")
        e2.record()
        CONV_TIMING.append(("wino_input_kernel", 0.0, e0, e1, shape, 0.0, "none", _stream()))
        CONV_TIMING.append((last_conv_variant(), 2.0 * B * Ho * Wo * p.cols * k_real, e1, e2, shape, 2.0 * tiles * 16 * p.cols * p.Cin,
                            {2: "bf16x6", 3: "f16x3"}.get(int(d.precision), "f32"), _stream()))
        return out
    e0, e1 = ev(), ev()
    e0.record()
    _lib.check(_lib.lib().a3d_conv2d_nhwc_f32(C.byref(d), _stream()), "a3d_conv2d_nhwc_f32")
    e1.record()
    # algorithmic FLOPs of a phase launch = its share (1/4) of the 3x3 conv over the upsampled tensor
    executed = 2.0 * B * Ho * Wo * p.cols * (4 * p.Cin if p.phase == 5 else k_real)  # (fused phases: 4 of the 9 taps per column)
    fl = 2.0 * B * Ho * Wo * p.cols * (9 * p.Cin) if p.phase else executed
    CONV_TIMING.append((last_conv_variant(), fl, e0, e1, shape, executed, {0: "f32", 1: "bf16", 2: "bf16x6", 3: "f16x3"}[int(d.precision)], _stream()))
    return out


WINO_MAX_HW = int(os.environ.get("A3D_WINO_MAX_HW", "0"))
WINO_TUNE = int(os.environ.get("A3D_WINO_TUNE", "0"))  # measurement knob: 23 | 24 = every fp16x2 Winograd GEMM in its lockstep | 64-tile form (the same bits)
UPS_FUSED = os.environ.get("A3D_UPS_FUSED", "1") != "0"  # (False: always the four-launch form; same bits)


def conv2d_ups(x: torch.Tensor, phases: Sequence[PackedConv], *, x2: Optional[torch.Tensor] = None, fused: Optional[bool] = None, tune: int = 0) -> torch.Tensor:
    """3x3 pad-1 conv over the nearest-x2 upsampling of (x || x2), as four source-grid 2x2 convs (pack_conv_ups_phases) -- in the
    default arithmetic as ONE launch over the 9 distinct taps (a3d_conv_desc.phase == 5).  Two kernels run that launch: the
    patch-resident one (csrc/conv_ph4p.hip, round 4, the default: the input patch of an 8 x 32 tile split once per chunk and held in
    LDS while the nine taps multiply it) and the tap-outer one (conv_x3w_kernel's PH4, tune 15), which agrees BIT FOR BIT with the
    four-launch form (test_fused_upsampled_conv_equals_the_four_phase_launches).  The patch-resident kernel reduces over (chunk, tap)
    instead of (tap, chunk): it agrees with the other two to fp32 rounding, so which form runs is a function of the layer and the
    arithmetic only, never of the batch.  Measured at 64 frames (tools/ups_bench.py; four launches | tap-outer | patch-resident, ms):
    8x10 0.107 | 0.078 | 0.060, 15x20 0.200 | 0.142 | 0.115, 30x40 0.382 | 0.403 | 0.400, 60x80 1.230 | 1.351 | 1.168, 120x160 -> 64
    channels 3.044 | 2.825 | 1.931."""
    B, H, W, _ = x.shape
    if AUDIT is not None and not AUDIT.busy and fused is None:
        return AUDIT.conv2d_ups(x, phases, x2)
    out = torch.empty((B, 2 * H, 2 * W, phases[0].cols), device=x.device, dtype=torch.float32)
    pinned = phases[0].pin_precision == 2  # (the audit pins the LAYER: it then runs bf16x3 -- since round 5 in the fused form as well)
    if fused is None:
        fused = UPS_FUSED and DEFAULT_PRECISION in (2, 3)  # (a function of the layer and the arithmetic: never of the batch)
    if fused:
        pf = getattr(phases[0], "_fused", None)
        if pf is None:
            pf = pack_conv_ups_fused(phases)
            phases[0]._fused = pf if pf is not None else False
        if pf:
            pf.name = getattr(phases[0], "name", "")  # (measurement tools label launches by layer: tools/kernel_table.py)
        b3 = pinned or DEFAULT_PRECISION == 2
        # (the bf16x3 launch of the fused form needs the pre-split filter, which _conv2d_launch builds from 192 columns on, and the phase-5
        # launcher's 2^31-byte filter bound: a narrower or oversized layer runs the four per-phase launches below, as before round 5)
        if pf and b3 and (pf.cols < 192 or pf.cols * pf.Kpad * 6 >= 2 ** 31):
            pf = None
        if pf:  # (tune 15 / 16: never / always the patch-resident kernel of csrc/conv_ph4p.hip; 0: by the map's size)
            return conv2d(x, pf, x2=x2, out=out, precision=2 if b3 else 3, tune=tune)
    for p in phases:
        conv2d(x, p, x2=x2, out=out)
    return out


DEPTH_PRED_FUSED = os.environ.get("A3D_DEPTH_PRED_FUSED", "1") != "0"


def conv2d_ups_to1(x: torch.Tensor, phases: Sequence[PackedConv], w9: torch.Tensor, bias: float, *, x2: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """conv3x3_to1(conv2d_ups(x, phases, x2), w9, bias) without the [B,2H,2W,64] tensor in between (a3d_conv_desc.dot_w / dot_y + a3d_tapsum9):
    the patch-resident four-phase launch stores, per output pixel, the nine dot products of its 64 channels with the nine taps' weights, and
    a second small launch adds the shifted planes.  w9 [3,3,64] (or [9,64]) fp32.  Returns [B,2H,2W], or None where the form does not
    apply (another arithmetic, a pinned layer, an audit pass, a batch past one launch): the caller runs the two layers.  Equal to them
    to fp32 rounding of the 576-term sums (another summation order)."""
    pinned = phases[0].pin_precision == 2
    if not (DEPTH_PRED_FUSED and UPS_FUSED and DEFAULT_PRECISION == 3 and not pinned and AUDIT is None and phases[0].cols == 64):
        return None
    B, H, W, C = x.shape
    pf = getattr(phases[0], "_fused", None)
    if pf is None:
        pf = pack_conv_ups_fused(phases)
        phases[0]._fused = pf if pf is not None else False
    if not pf:
        return None
    # What one launch addresses: the nine tap-product planes g [B,9,2H,2W] and the input(s) -- the [B,2H,2W,64] tensor is never
    # allocated.  A batch past the 32-bit offsets runs as blocks of images through the SAME two kernels (per-image arithmetic: the
    # bits of a frame do not depend on the block it travels in), never through another algorithm.
    per_image = max(9 * 4 * H * W * 4, H * W * C * 4, 0 if x2 is None else x2.shape[1] * x2.shape[2] * x2.shape[3] * 4)
    nb = max(1, min(B, _ADDR_LIMIT // per_image))
    y = torch.empty((B, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
    w9 = w9.reshape(9, 64)
    for b0 in range(0, B, nb):
        b1 = min(B, b0 + nb)
        cut = lambda t: t if (t is None or nb >= B) else keep_amax(t[b0:b1], _amax_rows(t, b0, b1))
        xb, x2b = cut(x), cut(x2)
        g = torch.empty((b1 - b0, 9, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
        conv2d(xb, pf, x2=x2b, out=g, precision=3, dot=(w9, g))
        _lib.check(_lib.lib().a3d_tapsum9(g.data_ptr(), float(bias), y[b0:b1].data_ptr(), b1 - b0, 2 * H, 2 * W, _stream()), "a3d_tapsum9")
    return y


_LAST_PRECISION = -1  # a3d_conv_desc.precision of the calling thread's last conv launch (what the mode rules resolved to)
AUDIT = None  # a PrecisionAudit while model.audit_precision() runs (offline, once per loaded checkpoint); None on the detection path


class PrecisionAudit:
    """Load-time check of the default arithmetic on the weights actually loaded.

    fp16x2 carries 22 significand bits under ONE power-of-two exponent per image: an output whose whole receptive field lies more than
    2^18 below its image's maximum loses bits ("what the format guarantees").  Random-init weights with calibrated
    batch norm never get there; a checkpoint nobody here has seen might.  While this object is installed (`with PrecisionAudit() as a:`)
    every conv / linear / deconv layer that the mode rules send to fp16x2 is ALSO evaluated in bf16x3 (exact 3-way splits: no block
    exponent, no window) on the same input, together with S = sum_k |x_k| |w_k| (the same layer on |x|, |w|), and every output element
    is held to the fp32-style ONE-term law at the layer's own scale:

        | y_fp16x2 - y_bf16x3 |  <=  1.25 c 2^-22 |scale_n| S,      c = 8 + sqrt(K) / 4      (tests/test_gpu_precision.py; Winograd layers:
                                                                     S max-pooled 3x3, the 4x4 input patch of a 2x2 output tile)

    (the epilogue -- folded BN scale, shift, residual, ReLU / LeakyReLU -- is 1-Lipschitz in the accumulator up to |scale_n|; 1.25 leaves
    room for the reference's own rounding).  A layer with any violating element is PINNED to bf16x3 (`PackedConv.pin_precision`, kept on
    its module): a static property of the layer, so a frame's result still does not depend on its batch.  The pass costs three launches
    per layer and a few torch reductions; it is not part of the detection path."""

    def __init__(self, pin: bool = True, slack: float = 1.25):
        self.pin, self.slack, self.busy = pin, slack, False
        self.rows = []  # dicts: layer, kernel, shape, K, max_ratio, violations, elements, pinned

    def __enter__(self):
        global AUDIT
        self._prev, AUDIT = AUDIT, self
        return self

    def __exit__(self, *exc):
        global AUDIT
        AUDIT = self._prev
        return False

    @staticmethod
    def _abs_pack(p: PackedConv) -> PackedConv:
        q = getattr(p, "_audit_abs", None)
        if q is None or q.w.data_ptr() == 0 or getattr(p, "_audit_abs_src", None) != (p.w.data_ptr(), p.w._version):
            q = PackedConv(p.w.abs(), None, None, p.KH, p.KW, p.stride, p.pad, p.Cin, p.cols, p.Kpad, ACT_NONE, p.pixshuf, p.stem, phase=p.phase)
            p._audit_abs, p._audit_abs_src = q, (p.w.data_ptr(), p.w._version)
        return q

    def _judge(self, p: PackedConv, variant: str, shape, y3, y2, S, K: int, wino: bool):
        c = 8.0 + math.sqrt(K) / 4.0
        if wino:  # a 2x2 output tile is computed from its whole 4x4 input patch
            S = torch.nn.functional.max_pool2d(S.permute(0, 3, 1, 2), 3, 1, 1).permute(0, 2, 3, 1)
        n = y3.shape[-1]
        sc = torch.ones(n, device=y3.device) if p.scale is None else p.scale[:n].abs()  # (per GEMM column = per output channel; deconvs carry none)
        bound = self.slack * c * 2.0 ** -22 * sc * S[..., :n] + 1e-37
        err = (y3 - y2).abs()
        finite = torch.isfinite(err) & torch.isfinite(bound)
        ratio = torch.where(finite, err / bound, torch.zeros_like(err))
        viol = int((ratio > 1.0).sum())
        row = dict(layer=p.name or f"<packed {id(p):x}>", kernel=variant, shape=shape, K=K, max_ratio=float(ratio.max()) if ratio.numel() else 0.0,
                   violations=viol, elements=int(ratio.numel()), pinned=False)
        if viol and self.pin:
            p.pin_precision = 2
            row["pinned"] = True
        self.rows.append(row)
        return row

    def conv2d(self, x, p: PackedConv, **kw):
        self.busy = True
        try:
            y = conv2d(x, p, **kw)
            variant, prec = last_conv_variant(), _LAST_PRECISION
            if prec != 3:
                return y
            kw2 = dict(kw)
            kw2["out"] = None
            y2 = conv2d(x, p, precision="bf16x3", **kw2)
            x2 = kw.get("x2")
            S = conv2d(x.abs(), self._abs_pack(p), x2=None if x2 is None else x2.abs(), ups=kw.get("ups", False), splitk=kw.get("splitk", 1),
                       m_dev=kw.get("m_dev"), precision="bf16x3", wino=False)
            live = None
            if kw.get("m_dev") is not None:  # ragged per-ROI batches: rows past the live count are not computed
                live = int(kw["m_dev"].reshape(-1)[0])
            yy, y2, S = (t if live is None else t[:live] for t in (y, y2, S))
            self._judge(p, variant, tuple(x.shape), yy, y2, S, 147 if p.stem else p.KH * p.KW * p.Cin, variant.startswith("wino"))
            return y
        finally:
            self.busy = False

    def conv2d_ups(self, x, phases, x2):
        self.busy = True
        try:
            y = conv2d_ups(x, phases, x2=x2)
            variant = last_conv_variant()
            if DEFAULT_PRECISION != 3 or phases[0].pin_precision == 2:
                return y
            out2 = torch.empty_like(y)
            S = torch.empty_like(y)
            for p in phases:  # the four phase launches: bf16x3 reference, and sum |x||w| per phase
                conv2d(x, p, x2=x2, out=out2, precision="bf16x3")
                conv2d(x.abs(), self._abs_pack(p), x2=None if x2 is None else x2.abs(), out=S, precision="bf16x3")
            row = self._judge(phases[0], variant, tuple(x.shape) + ("ups",), y, out2, S, 4 * phases[0].Cin, False)
            if row["pinned"]:
                for p in phases:
                    p.pin_precision = 2
            return y
        finally:
            self.busy = False

    def pinned(self):
        return [r for r in self.rows if r["pinned"]]


def linear(x: torch.Tensor, p: PackedConv, *, act: Optional[int] = None, splitk: int = 1,
           m_dev: Optional[torch.Tensor] = None, precision=None) -> torch.Tensor:
    """x [M, K] -> [M, cols]."""
    if x.dtype == torch.float16:  # pre-split rows [M, 1, 1, K/16, 2, 16] (roi_align_fpn(presplit=True), presplit_f16x2)
        M = x.shape[0]
        y = conv2d(x, p, act=act, splitk=splitk, m_dev=m_dev, precision=precision)
        return keep_amax(y.view(M, p.cols), y)
    M, K = x.shape
    y = conv2d(keep_amax(x.view(M, 1, 1, K), x), p, act=act, splitk=splitk, m_dev=m_dev, precision=precision)
    return keep_amax(y.view(M, p.cols), y)


def choose_splitk(M: int, cols: int, K: int) -> int:
    """Split-K factor of a linear layer.  Depends on K ONLY: the summation order of every output element must
    not change with the number of rows, otherwise a frame's result would depend on how frames were batched or
    sharded across GPUs.  The 50176-deep head FCs are weight-bandwidth bound at realistic row counts: 32 K-slices
    x 8 column tiles = 256 workgroups stream the 205 MB weight matrix once."""
    return 32 if K >= 16384 else 1


def _f3(v):
    return (C.c_float * 3)(*[float(t) for t in v])


def preprocess_u8hwc(frames: torch.Tensor, mean, std) -> torch.Tensor:
    _req(frames, torch.uint8)
    B, H, W, _ = frames.shape
    out = torch.empty((B, H, W, 4), device=frames.device, dtype=torch.float32)
    _lib.check(_lib.lib().a3d_preprocess_u8hwc(frames.data_ptr(), out.data_ptr(), B, H, W, _f3(mean), _f3(std), _stream()),
               "a3d_preprocess_u8hwc")
    _const_amax(out, max(max(abs(0.0 - m), abs(255.0 - m)) / sd for m, sd in zip(mean, std)))  # u8 input: a closed-form bound
    return out


def preprocess_resize_u8(frames: torch.Tensor, mean, std, out_hw=(480, 640), swap_rb: bool = True, want_u8: bool = False):
    """frames uint8 [B,Hs,Ws,3] in the reader's channel order (RGB from imageio) -> NHWC4 fp32 [B,Hd,Wd,4] normalised in the
    flipped order (BGR), as cv2.resize + [:, :, ::-1] + float + (x - mean)/std (tools/inference.py:216-218); optionally also
    the resized uint8 frames in the source order."""
    _req(frames, torch.uint8)
    B, Hs, Ws, _ = frames.shape
    Hd, Wd = int(out_hw[0]), int(out_hw[1])
    out = torch.empty((B, Hd, Wd, 4), device=frames.device, dtype=torch.float32)
    u8 = torch.empty((B, Hd, Wd, 3), device=frames.device, dtype=torch.uint8) if want_u8 else None
    _lib.check(_lib.lib().a3d_preprocess_resize_u8(frames.data_ptr(), out.data_ptr(), _p(u8), B, Hs, Ws, Hd, Wd, int(bool(swap_rb)),
                                                   _f3(mean), _f3(std), _stream()), "a3d_preprocess_resize_u8")
    _const_amax(out, max(max(abs(0.0 - m), abs(255.0 - m)) / sd for m, sd in zip(mean, std)))
    return (out, u8) if want_u8 else out


def preprocess_f32chw(images: torch.Tensor, mean, std) -> torch.Tensor:
    _req(images)
    B, _, H, W = images.shape
    out = torch.empty((B, H, W, 4), device=images.device, dtype=torch.float32)
    _lib.check(_lib.lib().a3d_preprocess_f32chw(images.data_ptr(), out.data_ptr(), B, H, W, _f3(mean), _f3(std), _stream()),
               "a3d_preprocess_f32chw")
    return out


STEM_POOL_FUSED = os.environ.get("A3D_STEM_POOL", "1") != "0"


def stem_pool(x: torch.Tensor, p: "PackedConv") -> Optional[torch.Tensor]:
    """BasicStem in ONE launch (a3d_stem_conv_pool): 7x7 s2 stem conv + folded BN + ReLU + 3x3 s2 max-pool of the fp16x2 arithmetic, bit
    for bit what conv2d(x, p) + maxpool3x3s2 give.  x [B,H,W,4] normalised.  Returns None where the fused kernel does not apply (another
    arithmetic, a pinned layer, an audit or timing pass that wants the two launches, tensors past the 32-bit limit): the caller then
    runs the two launches."""
    if not (STEM_POOL_FUSED and DEFAULT_PRECISION == 3 and p.stem and p.presplit and p.pin_precision != 2 and AUDIT is None and p.cols == 64):
        return None
    _req(x)
    B, H, W, _ = x.shape
    if B * H * W * 16 >= (1 << 31) or p.Kpad != 224 or p.w.shape[0] != 64:
        return None
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
    out = torch.empty((B, Hp, Wp, 64), device=x.device, dtype=torch.float32)
    d = _lib.ConvDesc()
    d.x, d.w, d.scale, d.shift, d.y = _p(x), _p(p.w), _p(p.scale), _p(p.shift), _p(out)
    d.B, d.H, d.W, d.Cin, d.Cin2 = B, H, W, 4, 0
    d.Ho, d.Wo, d.Cout = Ho, Wo, 64
    d.KH, d.KW, d.stride, d.pad = 7, 7, 2, 3
    d.Kpad, d.act, d.stem, d.splitk, d.precision = p.Kpad, p.act, 1, 1, 3
    d.tune = int(os.environ.get("A3D_STEM_ABL", "0"))  # (developer builds with -DA3D_ABLATIONS only: timing-only variants)
    d.in_amax = amax_of(x).data_ptr()
    if getattr(p, "_w_scale", None) is None:
        p._w_scale = _pow2_scale_host(float(p.w.abs().max()))
    d.w_scale = p._w_scale
    if p.w_h2 is None or p.w_h2.device != p.w.device:  # (the filter's fp16 planes: the cache ops.conv2d fills for every direct fp16x2 layer)
        p.w_h2 = torch.empty((p.Kpad // 16, 2, p.w.shape[0], 16), device=p.w.device, dtype=torch.float16)
        _lib.check(_lib.lib().a3d_split_f16x2_chunk(p.w.data_ptr(), p.w_h2.data_ptr(), 1, p.w.shape[0], p.Kpad, 16, d.w_scale, _stream()),
                   "a3d_split_f16x2_chunk")
        if not os.environ.get("A3D_NO_PUBLISH"):
            torch.cuda.current_stream().synchronize()
    d.w_x3 = p.w_h2.data_ptr()
    if not os.environ.get("A3D_NO_YAMAX"):
        out._a3d_amax = amax_slot(B, out.device)
        d.y_amax = out._a3d_amax.data_ptr()
    global _LAST_PRECISION
    _LAST_PRECISION = 3
    if CONV_TIMING is not None and CONV_TIMING_ONLY is None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(_lib.lib().a3d_stem_conv_pool(C.byref(d), _stream()), "a3d_stem_conv_pool")
        e1.record()
        fl = 2.0 * B * Ho * Wo * 64 * 147
        CONV_TIMING.append(("stem_pool_kernel", fl, e0, e1, f"{B}x{H}x{W}x4->64 k7 s2 + pool", fl, "f16x3", _stream()))
        return out
    _lib.check(_lib.lib().a3d_stem_conv_pool(C.byref(d), _stream()), "a3d_stem_conv_pool")
    return out


def maxpool3x3s2(x: torch.Tensor) -> torch.Tensor:
    _req(x)
    B, H, W, Cc = x.shape
    out = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().a3d_maxpool3x3s2_nhwc(x.data_ptr(), out.data_ptr(), B, H, W, Cc, _stream()), "a3d_maxpool3x3s2_nhwc")
    keep_amax(out, x)  # (a max / a copy / a convex combination of x's values: x's per-image maxima bound out's)
    return out


def subsample2(x: torch.Tensor) -> torch.Tensor:
    _req(x)
    B, H, W, Cc = x.shape
    out = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().a3d_subsample2_nhwc(x.data_ptr(), out.data_ptr(), B, H, W, Cc, _stream()), "a3d_subsample2_nhwc")
    keep_amax(out, x)  # (a max / a copy / a convex combination of x's values: x's per-image maxima bound out's)
    return out


def resize_bilinear(x: torch.Tensor, Ho: int, Wo: int) -> torch.Tensor:
    _req(x)
    B, H, W, Cc = x.shape
    out = torch.empty((B, Ho, Wo, Cc), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().a3d_resize_bilinear_nhwc(x.data_ptr(), out.data_ptr(), B, H, W, Cc, Ho, Wo, _stream()),
               "a3d_resize_bilinear_nhwc")
    keep_amax(out, x)  # (a max / a copy / a convex combination of x's values: x's per-image maxima bound out's)
    return out


def conv3x3_to1(x: torch.Tensor, w: torch.Tensor, bias: float) -> torch.Tensor:
    _req(x)
    _req(w)
    B, H, W, Cc = x.shape
    out = torch.empty((B, H, W), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().a3d_conv3x3_to1_nhwc(x.data_ptr(), w.data_ptr(), float(bias), out.data_ptr(), B, H, W, Cc, _stream()),
               "a3d_conv3x3_to1_nhwc")
    return out


def group_workspace(n_groups: int, device) -> torch.Tensor:
    return torch.empty(_lib.lib().a3d_group_buffers_bytes(n_groups), device=device, dtype=torch.uint8)


def carve_group_workspace(ws: torch.Tensor, G: int) -> dict:
    """Views of the selection workspace (layout of `carve()` in csrc/proposals.hip): per group the
    score-sorted candidate boxes, scores, tie-break positions, validity and NMS keep flags.  Test hook."""
    cap = GROUP_CAP
    o = 0
    out = {}
    for name, width, dt in (("boxes", 16, torch.float32), ("scores", 4, torch.float32), ("pos", 4, torch.int32),
                            ("valid", 4, torch.int32), ("keep", 4, torch.int32)):
        nbytes = G * cap * width
        t = ws[o:o + nbytes].view(dt)
        out[name] = t.view(G, cap, 4) if name == "boxes" else t.view(G, cap)
        o += nbytes
    out["n"] = ws[o:o + G * 4].view(torch.int32)
    return out


def rpn_proposals(heads: Sequence[torch.Tensor], strides: Sequence[int], cell_anchors: torch.Tensor, img_hw, *,
                  pre_topk: int, post_topk: int, nms_thresh: float, min_size: float, weights, scale_clamp: float,
                  return_groups: bool = False):
    """heads[l]: [B,Hf,Wf,CH] (objectness 0..A-1, deltas A..5A-1).  cell_anchors: CPU float32 [L,3,4].
    -> boxes [B,post,4], logits [B,post], level [B,post] int32, pos [B,post] int32, count [B] int32."""
    L = len(heads)
    B = heads[0].shape[0]
    dev = heads[0].device
    d = _lib.RpnDesc()
    for l, h in enumerate(heads):
        _req(h)
        d.head[l] = h.data_ptr()
        d.Hf[l], d.Wf[l], d.stride[l] = h.shape[1], h.shape[2], int(strides[l])
        for a in range(3):
            for j in range(4):
                d.cell_anchors[l][a][j] = float(cell_anchors[l, a, j])
    d.B, d.L, d.A, d.CH = B, L, 3, heads[0].shape[3]
    d.img_h, d.img_w = int(img_hw[0]), int(img_hw[1])
    d.pre_topk, d.post_topk = int(pre_topk), int(post_topk)
    d.nms_thresh, d.min_size = float(nms_thresh), float(min_size)
    for j in range(4):
        d.weights[j] = float(weights[j])
    d.scale_clamp = float(scale_clamp)
    if pre_topk <= GROUP_CAP:
        ws = group_workspace(B * L, dev)
    else:  # training's 2000 candidates per level: 2048-slot groups + global NMS words
        assert not return_groups, "group buffers are only exposed in the 1024-slot layout"
        ws = torch.empty(_lib.lib().a3d_rpn_workspace_bytes(B, L, int(pre_topk)), device=dev, dtype=torch.uint8)
    boxes = torch.empty((B, post_topk, 4), device=dev, dtype=torch.float32)
    scores = torch.empty((B, post_topk), device=dev, dtype=torch.float32)
    level = torch.empty((B, post_topk), device=dev, dtype=torch.int32)
    pos = torch.empty((B, post_topk), device=dev, dtype=torch.int32)
    count = torch.empty((B,), device=dev, dtype=torch.int32)
    d.workspace = ws.data_ptr()
    d.out_boxes, d.out_scores, d.out_level, d.out_pos, d.out_count = (boxes.data_ptr(), scores.data_ptr(), level.data_ptr(),
                                                                      pos.data_ptr(), count.data_ptr())
    _lib.check(_lib.lib().a3d_rpn_proposals(C.byref(d), _stream()), "a3d_rpn_proposals")
    if return_groups:
        return boxes, scores, level, pos, count, carve_group_workspace(ws, B * L)
    return boxes, scores, level, pos, count


def box_detections(pred: torch.Tensor, prop_boxes: torch.Tensor, prop_count: torch.Tensor, img_hw, *, num_classes: int,
                   score_thresh: float, nms_thresh: float, topk: int, weights, scale_clamp: float,
                   return_groups: bool = False):
    """pred [B*R, CH] (cls logits 0..C, deltas C+1..), prop_boxes [B,R,4], prop_count [B] int32."""
    _req(pred)
    _req(prop_boxes)
    _req(prop_count, torch.int32)
    B, R, _ = prop_boxes.shape
    dev = pred.device
    d = _lib.BoxDetDesc()
    d.pred, d.prop_boxes, d.prop_count = pred.data_ptr(), prop_boxes.data_ptr(), prop_count.data_ptr()
    d.B, d.R, d.C, d.CH = B, R, int(num_classes), pred.shape[1]
    d.img_h, d.img_w = int(img_hw[0]), int(img_hw[1])
    d.score_thresh, d.nms_thresh, d.topk = float(score_thresh), float(nms_thresh), int(topk)
    for j in range(4):
        d.weights[j] = float(weights[j])
    d.scale_clamp = float(scale_clamp)
    ws = group_workspace(B * num_classes, dev)
    boxes = torch.empty((B, topk, 4), device=dev, dtype=torch.float32)
    scores = torch.empty((B, topk), device=dev, dtype=torch.float32)
    classes = torch.empty((B, topk), device=dev, dtype=torch.int32)
    pos = torch.empty((B, topk), device=dev, dtype=torch.int32)
    count = torch.empty((B,), device=dev, dtype=torch.int32)
    d.workspace = ws.data_ptr()
    d.out_boxes, d.out_scores, d.out_classes, d.out_pos, d.out_count = (boxes.data_ptr(), scores.data_ptr(),
                                                                        classes.data_ptr(), pos.data_ptr(), count.data_ptr())
    _lib.check(_lib.lib().a3d_box_detections(C.byref(d), _stream()), "a3d_box_detections")
    if return_groups:
        return boxes, scores, classes, pos, count, carve_group_workspace(ws, B * num_classes)
    return boxes, scores, classes, pos, count


def group_nms(g_boxes: torch.Tensor, g_valid: torch.Tensor, g_n: torch.Tensor, thresh: float) -> torch.Tensor:
    """g_boxes [G,1024,4] score-descending, g_valid [G,1024] int32, g_n [G] int32 -> keep [G,1024] int32."""
    _req(g_boxes)
    _req(g_valid, torch.int32)
    _req(g_n, torch.int32)
    G = g_boxes.shape[0]
    assert g_boxes.shape[1] == GROUP_CAP
    keep = torch.empty((G, GROUP_CAP), device=g_boxes.device, dtype=torch.int32)
    _lib.check(_lib.lib().a3d_group_nms(g_boxes.data_ptr(), g_valid.data_ptr(), g_n.data_ptr(), keep.data_ptr(), G,
                                        float(thresh), _stream()), "a3d_group_nms")
    return keep


ROI_SERIAL = False  # a3d_roialign_desc.serial (schedule only: one cell load at a time, the form the batched loads replaced)
ROI_SPATIAL_ORDER = True  # walk every image's boxes in (level, y, x) order (schedule only: same output rows, same bits)
# The 7x7 box pooler's rolling-window walk (a3d_roialign_desc.serial == 2; csrc/roi_align.hip): a wave owns two bin rows and walks the ROI's
# cell columns once -- 675 instead of 980 cell loads per typical ROI, 2.04 ms against 2.43 at 64 x 1000 boxes.  Sums in (column, row) order:
# equal to the bin-by-bin walk to fp32 rounding (3e-7), a function of the ROI alone.  "0": the bin-by-bin walk (the round-5 bits).
ROI_ROLLING = os.environ.get("A3D_ROI_ROLLING", "1") != "0"


def roi_align_fpn(feats: Sequence[torch.Tensor], scales: Sequence[float], boxes: torch.Tensor,
                  count: Optional[torch.Tensor], P: int, sampling_ratio: int, aligned: bool, *,
                  row_offset: Optional[torch.Tensor] = None, rows: Optional[int] = None, want_level: bool = False,
                  zero: bool = False, presplit: bool = False):
    """feats[l] NHWC [B,Hf,Wf,C]; boxes [B,R,4]; -> [rows, P, P, C] (rows = B*R unless compacted).

    presplit (default arithmetic only; a3d_roialign_desc.out_h2): the pooled rows leave the kernel as the two scaled fp16 planes the
    fp16x2 GEMM multiplies -- a torch.float16 tensor [rows, 1, 1, P*P*C/16, 2, 16] (a3d_conv_desc.x_h2) carrying the per-row maxima;
    `conv2d` / `linear` take it in place of the fp32 rows and move both operands by LDS-DMA."""
    _req(boxes)
    B, R, _ = boxes.shape
    Cc = feats[0].shape[3]
    dev = boxes.device
    nrows = B * R if rows is None else rows
    # rows of slots past count[b] are never written NOR read downstream (GEMM rows are independent and the selection
    # kernels only look at live rows), so the buffer is not cleared (a 3 GB memset per 64-frame step otherwise)
    presplit = bool(presplit) and DEFAULT_PRECISION == 3 and Cc == 256 and P * P * Cc * 4 <= 120 * 1024
    if presplit:
        out = (torch.zeros if zero else torch.empty)((nrows, 1, 1, P * P * Cc // 16, 2, 16), device=dev, dtype=torch.float16)
    else:
        out = (torch.zeros if zero else torch.empty)((nrows, P, P, Cc), device=dev, dtype=torch.float32)
    lvl = torch.full((nrows,), -1, device=dev, dtype=torch.int32) if want_level else None
    d = _lib.RoiAlignDesc()
    for l, f in enumerate(feats):
        _req(f)
        d.feat[l] = f.data_ptr()
        d.Hf[l], d.Wf[l] = f.shape[1], f.shape[2]
        d.scale[l] = float(scales[l])
    d.L, d.C = len(feats), Cc
    d.boxes, d.count, d.row_offset = boxes.data_ptr(), _p(count), _p(row_offset)
    d.B, d.R, d.P, d.sampling_ratio, d.aligned = B, R, int(P), int(sampling_ratio), int(bool(aligned))
    d.out, d.out_level = (None if presplit else out.data_ptr()), _p(lvl)
    if presplit:
        d.out_h2 = out.data_ptr()
    # (measured, tools/roi_bench.py: -7 % on the 1000-proposal box pooler; the 100-detection poolers lose 3-5 % to the sort launch)
    order = torch.empty((B * R,), device=dev, dtype=torch.int32) if (512 <= R <= 1024 and ROI_SPATIAL_ORDER) else None
    d.order_ws = _p(order)
    d.serial = 1 if ROI_SERIAL else (2 if (ROI_ROLLING and P == 7 and not presplit) else 0)  # (1: schedule only -- the bit-equality test and tools/roi_bench.py set it)
    ra = None
    if DEFAULT_PRECISION == 3:  # fp16x2: every pooled row records ITS OWN maximum (the scale of the layers that consume it) ...
        ra = amax_slot(nrows, dev)
        d.out_amax = ra.data_ptr()
        for l, f in enumerate(feats):  # ... and ROIs fainter than 2^-16 of their level are counted (roi_window_count)
            d.level_amax[l] = amax_of(f).data_ptr()
        d.window_count = window_counter(dev).data_ptr()
    _lib.check(_lib.lib().a3d_roi_align_fpn(C.byref(d), _stream()), "a3d_roi_align_fpn")
    if ra is not None:
        out._a3d_amax = ra
    return (out, lvl) if want_level else out


_WINDOW_COUNTERS: dict = {}


def window_counter(device) -> torch.Tensor:
    """Device-side int32 counter of the default arithmetic's window monitor (a3d_roialign_desc.window_count): ROIs whose own maximum
    lies below 2^-16 of their pyramid level's, i.e. whose pooled features the per-image block exponents of the backbone no longer
    resolve to fp32's own rounding.  Accumulates over the process; `roi_window_count()` reads it (one synchronisation)."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    t = _WINDOW_COUNTERS.get(device)
    if t is None:
        t = _WINDOW_COUNTERS[device] = torch.zeros(1, device=device, dtype=torch.int32)
        torch.cuda.current_stream(device).synchronize()  # (visible to every stream that will count into it)
    return t


def roi_window_count(device="cuda") -> int:
    return int(window_counter(device).item())


def count_offsets(count: torch.Tensor, cap: int) -> torch.Tensor:
    _req(count, torch.int32)
    B = count.shape[0]
    off = torch.empty((B + 1,), device=count.device, dtype=torch.int32)
    _lib.check(_lib.lib().a3d_count_offsets(count.data_ptr(), off.data_ptr(), B, int(cap), _stream()), "a3d_count_offsets")
    return off


def linear_small(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], norm_n: int = 0, sigmoid: bool = False,
                 m_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    _req(x)
    _req(w)
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().a3d_linear_small(x.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), M, _p(m_dev), K, N, int(norm_n),
                                           int(sigmoid), _stream()), "a3d_linear_small")
    return y


def paste_lsq(boxes, scores, count, row_offset, mask_prob, normals, depth, img_hw, *, post_score_thresh=0.1,
              mask_thresh=0.5, focal=571.623718, cx=319.5, cy=239.5, want_masks=True, clip_boxes=True):
    _req(boxes)
    _req(scores)
    B, R, _ = boxes.shape
    H, W = int(img_hw[0]), int(img_hw[1])
    dev = boxes.device
    MS = mask_prob.shape[-1]
    masks = torch.empty((B, R, H, W), device=dev, dtype=torch.uint8) if want_masks else None
    planes = torch.empty((B, R, 3), device=dev, dtype=torch.float32)
    area = torch.empty((B, R), device=dev, dtype=torch.int32)
    keep = torch.empty((B, R), device=dev, dtype=torch.int32)
    out_boxes = torch.empty((B, R, 4), device=dev, dtype=torch.float32)
    d = _lib.PasteDesc()
    d.boxes, d.scores, d.count, d.row_offset = boxes.data_ptr(), scores.data_ptr(), count.data_ptr(), row_offset.data_ptr()
    d.mask_prob, d.normals, d.depth = mask_prob.data_ptr(), _p(normals), _p(depth)
    d.B, d.R, d.MS, d.H, d.W = B, R, MS, H, W
    d.post_score_thresh, d.mask_thresh = float(post_score_thresh), float(mask_thresh)
    d.focal, d.cx, d.cy = float(focal), float(cx), float(cy)
    d.clip_boxes = int(bool(clip_boxes))
    d.masks, d.planes, d.area, d.keep, d.out_boxes = _p(masks), planes.data_ptr(), area.data_ptr(), keep.data_ptr(), out_boxes.data_ptr()
    _lib.check(_lib.lib().a3d_paste_lsq(C.byref(d), _stream()), "a3d_paste_lsq")
    return masks, planes, area, keep, out_boxes


def record_floats(MS: int = 28) -> int:
    return int(_lib.lib().a3d_record_floats(MS))


def detections_pack(boxes, scores, classes, count, row_offset, keep, planes, rot_axis, tran_axis, mask_prob, MS=28):
    B, R, _ = boxes.shape
    dev = boxes.device
    rec = record_floats(MS)
    records = torch.empty((B, R, rec), device=dev, dtype=torch.float32)
    rec_count = torch.empty((B,), device=dev, dtype=torch.int32)
    d = _lib.PackDesc()
    d.boxes, d.scores, d.classes, d.count = boxes.data_ptr(), scores.data_ptr(), classes.data_ptr(), count.data_ptr()
    d.row_offset, d.keep = row_offset.data_ptr(), keep.data_ptr()
    d.planes, d.rot_axis, d.tran_axis, d.mask_prob = _p(planes), _p(rot_axis), _p(tran_axis), _p(mask_prob)
    d.B, d.R, d.MS = B, R, MS
    d.records, d.rec_count = records.data_ptr(), rec_count.data_ptr()
    _lib.check(_lib.lib().a3d_detections_pack(C.byref(d), _stream()), "a3d_detections_pack")
    return records, rec_count
