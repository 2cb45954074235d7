"""Synthetic workload helpers: seeded frames and batch-norm calibration for random-init weights.

The reference ships no weights offline (`exps/model_final.pth`, config.yaml:312) and with detectron2's
initialisers a random-init ResNet-50 with identity FrozenBN overflows to ~1e4 activations by res5, so the
RPN emits zero proposals and the detector is degenerate.  `calibrate_batchnorm` gives every frozen / eval
batch-norm the running statistics of its own input on a calibration batch (run through the HIP path
itself), which is what a trained checkpoint's buffers contain; the detector then yields ~1000 proposals per
frame and a realistic score spread."""
from __future__ import annotations

import numpy as np
import torch

from ..modeling.layers import _CalibrationState


def synthetic_frames(n: int, seed: int = 2020, h: int = 480, w: int = 640) -> np.ndarray:
    """uint8 uniform[0,255] BGR frames (n,h,w,3); 2020 is the reference's seed (tools/inference.py:172-173)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)


@torch.no_grad()
def calibrate_batchnorm(model, frames_u8: torch.Tensor) -> None:
    """One pass of `frames_u8` (uint8 [B,H,W,3], device) through backbone + depth head with every norm
    layer re-estimating its running statistics."""
    assert not model.training
    _CalibrationState.active = True
    try:
        from .. import ops

        x4 = ops.preprocess_u8hwc(frames_u8.contiguous(), model.pixel_mean, model.pixel_std)
        feats = model.backbone.forward_nhwc(x4)
        if getattr(model, "depth_head_on", False):
            model.depth_head.forward_nhwc(feats)
    finally:
        _CalibrationState.active = False
    torch.cuda.synchronize()
