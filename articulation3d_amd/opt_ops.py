"""Host-side wrappers over the optimiser-sweep kernels of include/a3d.h: bit-packed masks,
hypothesis projection, IoU matrix.  torch tensors carry device memory only."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch

from . import _lib
from .ops import _p, _req, _stream

MAX_HYP = 64


def words_per_mask(H: int, W: int) -> int:
    return (H * W + 31) // 32


def pack_masks(masks_u8: torch.Tensor) -> torch.Tensor:
    """[n,H,W] uint8 (non-zero = set) -> [n, words] int32 bit masks."""
    n, H, W = _req(masks_u8, torch.uint8).shape
    bits = torch.empty((n, words_per_mask(H, W)), device=masks_u8.device, dtype=torch.int32)
    _lib.check(_lib.lib().a3d_masks_pack_bits(_p(masks_u8), _p(bits), n, H, W, _stream()), "a3d_masks_pack_bits")
    return bits


def unpack_masks(bits: torch.Tensor, H: int, W: int) -> torch.Tensor:
    n = _req(bits, torch.int32).shape[0]
    out = torch.empty((n, H, W), device=bits.device, dtype=torch.uint8)
    _lib.check(_lib.lib().a3d_masks_unpack_bits(_p(bits), _p(out), n, H, W, _stream()), "a3d_masks_unpack_bits")
    return out


def project_hypotheses(mask_u8: torch.Tensor, normal: Sequence[float], offset: float, pivot: Sequence[float], xforms: torch.Tensor, *,
                       focal: float, cx: float, cy: float) -> torch.Tensor:
    """mask [H,W] uint8 on the device, xforms [A,12] fp32 (R row-major | t) -> bit masks [A, words] of the A re-projections."""
    H, W = _req(mask_u8, torch.uint8).shape
    A = _req(xforms).shape[0]
    assert xforms.shape[1] == 12 and A <= MAX_HYP
    d = _lib.SweepDesc()
    d.mask, d.H, d.W = _p(mask_u8), H, W
    for i in range(3):
        d.normal[i], d.pivot[i] = float(normal[i]), float(pivot[i])
    d.offset, d.focal, d.cx, d.cy = float(offset), float(focal), float(cx), float(cy)
    out = torch.empty((A, words_per_mask(H, W)), device=mask_u8.device, dtype=torch.int32)
    d.xforms, d.A, d.out_bits = _p(xforms), A, _p(out)
    _lib.check(_lib.lib().a3d_project_hypotheses(C.byref(d), _stream()), "a3d_project_hypotheses")
    return out


def mask_iou_matrix(target_bits: torch.Tensor, proj_bits: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """[F,words] x [A,words] -> IoU [F,A] fp32."""
    F_, A = _req(target_bits, torch.int32).shape[0], _req(proj_bits, torch.int32).shape[0]
    iou = torch.empty((F_, A), device=target_bits.device, dtype=torch.float32)
    _lib.check(_lib.lib().a3d_mask_iou_matrix(_p(target_bits), _p(proj_bits), _p(iou), F_, A, H, W, _stream()), "a3d_mask_iou_matrix")
    return iou
