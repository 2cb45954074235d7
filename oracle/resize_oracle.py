"""CPU restatement of `cv2.resize(img, (Wd, Hd))` on uint8 images with the default interpolation (INTER_LINEAR), the first step
of the reference's frame loop (/root/reference/articulation3d/tools/inference.py:216), plus the BGR flip (:218) and the
float cast / normalisation that follow it (pkg/utils/arti_vis.py:58, pkg/modeling/meta_arch/planercnn.py:188-196).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED: OpenCV is a third-party dependency that is neither vendored under /root/reference nor installed in this image
(the reference pins no version: README.md:55 `pip install ... opencv-python`), and the reference holds no test or golden
vector for this step.  The algorithm below is OpenCV's published one (modules/imgproc/src/resize.cpp, unchanged across 3.x /
4.x for 8-bit INTER_LINEAR):

  * equal sizes: copy;
  * an exact 2x decimation in both directions: INTER_LINEAR is switched to the fast INTER_AREA path,
    dst = (s00 + s01 + s10 + s11 + 2) >> 2;
  * otherwise fixed-point bilinear: fx = (float)((dx + 0.5) * (Ws / Wd) - 0.5), sx = floor(fx), fx -= sx, clamped to the image with
    fx = 0 at the borders; coefficients cvRound(w * 2048) as int16 (round half to even); horizontal pass in int32
    (r = s[sx] * a0 + s[sx + 1] * a1), vertical pass dst = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.
"""
from __future__ import annotations

import numpy as np


def _coeffs(dst: int, src: int):
    scale = np.float64(src) / np.float64(dst)
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int32)  # cvRound: half to even
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int32)
    return s, np.minimum(s + 1, src - 1), a0, a1


def cv2_resize_linear_u8(img: np.ndarray, dsize) -> np.ndarray:
    """img uint8 [Hs, Ws, C]; dsize = (Wd, Hd) as cv2 takes it.  -> uint8 [Hd, Wd, C]."""
    assert img.dtype == np.uint8 and img.ndim == 3
    Wd, Hd = int(dsize[0]), int(dsize[1])
    Hs, Ws = img.shape[:2]
    if (Hs, Ws) == (Hd, Wd):
        return img.copy()
    s = img.astype(np.int32)
    if Hs == 2 * Hd and Ws == 2 * Wd:
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, sx1, a0, a1 = _coeffs(Wd, Ws)
    sy, sy1, b0, b1 = _coeffs(Hd, Hs)
    rows = s[:, sx] * a0[None, :, None] + s[:, sx1] * a1[None, :, None]  # [Hs, Wd, C] int32
    r0, r1 = rows[sy], rows[sy1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def frontend(frames_rgb: np.ndarray, mean, std, out_hw=(480, 640)):
    """The reference's per-frame input steps on a clip: resize, keep the RGB frame, flip to BGR, float, normalise.
    -> (x [F, Hd, Wd, 3] float32 normalised BGR, resized uint8 RGB frames)."""
    res = np.stack([cv2_resize_linear_u8(f, (out_hw[1], out_hw[0])) for f in frames_rgb])
    bgr = res[..., ::-1].astype(np.float32)
    x = (bgr - np.asarray(mean, dtype=np.float32)) / np.asarray(std, dtype=np.float32)
    return x, res
