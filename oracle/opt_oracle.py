"""CPU oracle of the temporal optimiser's hypothesis sweeps -- TEST INFRASTRUCTURE ONLY.

Restates, in numpy, the numerical blocks of the reference's optimize_planes('3dc')
(articulation3d/articulation3d/utils/opt_utils.py:382-683 rotation clustering, :685-959 translation clustering,
:962-975 dispatcher) and the helpers they call:
  get_pcd / project2D            articulation3d/utils/vis.py:62-102           (focal 517.97, principal point (W/2, H/2))
  angle_offset_to_axis           articulation3d/data/planercnn_transforms.py:101-176 (with get_boundary_point)
  axis_to_angle_offset           articulation3d/data/planercnn_transforms.py:31-68
pytorch3d (Transform3d, Rotate, axis_angle_to_matrix) is a third-party dependency that is NOT vendored under
/root/reference (README install recipe, unpinned) -- its published algorithm is restated: axis-angle -> quaternion ->
matrix, and Rotate applies points @ R (row vectors), i.e. the TRANSPOSE of the column-vector rotation matrix.
PINNED (round 5) for the helpers that are the reference's own pure functions: get_pcd, project2D, axis_to_angle_offset,
angle_offset_to_axis and get_boundary_point are checked against the outputs of the reference's modules themselves
(oracle/make_golden.py sections 5-6 import data/planercnn_transforms.py and utils/vis.py behind name-only placeholders for
their unused imports -> tests/golden/axis_transforms.npz, pcd_project.npz; tests/test_oracle_golden.py).  The pin found one
deviation: the vertical-line sentinel of get_boundary_point is the float64 constant -pi/2, which only angle_offset_to_axis's
own `sin == 0` branch produces -- a float32 arctan never equals it (fixed here and in the product's host-side mirror).
PARITY UNPINNED for the rest -- the clustering / regression control flow of optimize_planes (the reference holds no test or
vector for it and its module imports pytorch3d at scope) and the pytorch3d rotation, restated from its published algorithm:
for those this file is the parity definition.
The projected pixel index is a float -> long truncation, so two fp32 implementations can disagree on the few points
that land within rounding of a pixel boundary; masks are compared by differing-pixel count and IoU, not bit for bit.
"""
from __future__ import annotations

import math
import random
from typing import Dict, List

import numpy as np
from scipy.stats import linregress

FOCAL, IMG_H, IMG_W = 517.97, 480, 640


def get_pcd(verts, normal, offset, h=IMG_H, w=IMG_W, focal=FOCAL):
    """vis.py:86-102 in float64: depth = offset / (n . K^-1 q), point = depth * K^-1 q.
    K^-1 q is written out ((x - w/2) / f, (y - h/2) / f, 1) and the dot product is summed left to right: the reference's
    np.linalg.inv / matrix products leave the summation order to LAPACK / BLAS, the parity definition fixes it."""
    v = np.asarray(verts, dtype=np.float64)
    rx, ry = (v[:, 0] - w / 2) / focal, (v[:, 1] - h / 2) / focal
    n = np.asarray(normal, dtype=np.float32).astype(np.float64)
    depth = np.float64(np.float32(offset)) / ((n[0] * rx + n[1] * ry) + n[2])
    return np.stack([depth * rx, depth * ry, depth], 1)


def project2d(pcd32, h=IMG_H, w=IMG_W, focal=FOCAL):
    """vis.py:62-76 (the fp32 tensor branch): K @ p, divide by z -- in float32, one rounding per operation,
    u = (f*x + (w/2)*z) / z (the zero products of K's empty entries are dropped; the order of the rest is fixed here)."""
    f, cx, cy = np.float32(focal), np.float32(w / 2), np.float32(h / 2)
    x, y, z = pcd32[:, 0], pcd32[:, 1], pcd32[:, 2]
    return np.stack([(f * x + cx * z) / z, (f * y + cy * z) / z], 1)


def axis_angle_to_matrix(axis_angle):
    """pytorch3d.transforms.axis_angle_to_matrix (via quaternions), float64 in / out."""
    aa = np.asarray(axis_angle, dtype=np.float64)
    angles = np.linalg.norm(aa, axis=-1, keepdims=True)
    half = 0.5 * angles
    small = np.abs(angles) < 1e-6
    s = np.where(small, 0.5 - angles * angles / 48, np.sin(half) / np.where(small, 1.0, angles))
    q = np.concatenate([np.cos(half), aa * s], -1)
    r, i, j, k = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    two_s = 2.0 / (q * q).sum(-1)
    o = np.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                  two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                  two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def get_boundary_point(y, x, angle, H, W):
    """planercnn_transforms.py:131-176: the two points where the line through (x, y) with slope tan(angle) meets the image."""
    p1 = p2 = None
    if float(angle) == -np.pi / 2:  # (the reference compares with the float64 constant: only ITS OWN sentinel for sin == 0 matches, never a float32 arctan)
        return (x, 0), (x, H - 1)
    if angle == 0.0:
        return (0, y), (W - 1, y)
    k = np.tan(angle)

    def push(pt):
        nonlocal p1, p2
        if p1 is None:
            p1 = pt
        elif p2 is None:
            p2 = pt
            if p2 == p1:
                p2 = None

    if 0 <= y - k * x < H:
        push((0, int(y - k * x)))
    if 0 <= k * (W - 1) + y - k * x < H:
        push((W - 1, int(k * (W - 1) + y - k * x)))
    if 0 <= x - y / k < W:
        push((int(x - y / k), 0))
    if 0 <= x - y / k + (H - 1) / k < W:
        push((int(x - y / k + (H - 1) / k), H - 1))
    if p2 is None:
        p2 = p1
    return p1, p2


def angle_offset_to_axis(angle_offsets, centers, H=IMG_H, W=IMG_W):
    """planercnn_transforms.py:101-129, fp32 scalars as in the reference -> int64 [n,4] (x1,y1,x2,y2)."""
    out = []
    for ao, c in zip(np.asarray(angle_offsets, dtype=np.float32), np.asarray(centers, dtype=np.float32)):
        sin, cos, p = ao[0], ao[1], np.float32(ao[2] * np.float32(100))
        x0, y0 = c
        angle = -np.pi / 2 if sin == 0 else np.float32(-np.arctan(cos / sin))
        x, y = np.float32(p * cos + x0), np.float32(p * sin + y0)
        p1, p2 = get_boundary_point(y, x, angle, H, W)
        out.append([0, 0, 1, 1] if p1 is None else [p1[0], p1[1], p2[0], p2[1]])
    return np.asarray(out).astype(np.int64)


def axis_to_angle_offset(axis, center):
    """planercnn_transforms.py:31-68 (mine=False): [n,4] pixel axis + [n,2] centre -> [n,4] (sin, cos, offset/100, valid)."""
    a = np.asarray(axis, dtype=np.float32) - np.concatenate([center, center], 1).astype(np.float32)
    x1, y1, x2, y2 = a[:, 0:1], a[:, 1:2], a[:, 2:3], a[:, 3:4]
    A, B, C = y1 - y2, x2 - x1, x1 * y2 - x2 * y1
    lll = np.sqrt(A * A + B * B)
    off = np.abs(C) / lll / np.float32(100)
    cos, sin = -A * np.sign(C) / lll, -B * np.sign(C) / lll
    return np.concatenate([sin, cos, off, np.ones_like(off)], 1).astype(np.float32)


def swap_plane(plane):
    """(a, b, c) -> (a, -c, b): opt_utils.py:402-404."""
    p = np.asarray(plane, dtype=np.float32).copy()
    return np.array([p[0], -p[2], p[1]], dtype=np.float32)


def plane_geometry(mask, plane, axis_pts):
    """opt_utils.py:401-415: unit normal, offset, the two 3-D axis points, unit direction and the fp32 point cloud."""
    pl = swap_plane(plane)
    offset = np.float32(np.linalg.norm(pl))
    normal = (pl / max(offset, np.float32(1e-12))).astype(np.float32)
    ys, xs = np.nonzero(np.asarray(mask) > 0)
    verts = np.stack([xs, ys], 1)  # nonzero().flip(1): (x, y)
    axis3d = get_pcd(np.asarray(axis_pts).reshape(-1, 2), normal, offset)
    d = axis3d[1] - axis3d[0]
    d = d / np.linalg.norm(d)
    pcd = get_pcd(verts, normal, offset).astype(np.float32)
    return normal, offset, axis3d, d, pcd


def rotation_hypotheses(angles32, dir_vec, pivot):
    """Transform3d.translate(p0).inverse -> Rotate(R) -> translate(p0) with points @ R (row vectors): x' = R^T (x - p0) + p0."""
    R = axis_angle_to_matrix(np.asarray(angles32, dtype=np.float32).astype(np.float64)[:, None] * dir_vec[None, :]).astype(np.float32)
    return [(np.ascontiguousarray(R[i].T), np.zeros(3, np.float32)) for i in range(len(R))], np.asarray(pivot, dtype=np.float32)


def translation_hypotheses(steps32, dir_vec):
    t = (np.asarray(steps32, dtype=np.float32).astype(np.float64)[:, None] * dir_vec[None, :]).astype(np.float32)
    return [(np.eye(3, dtype=np.float32), t[i]) for i in range(len(t))], np.zeros(3, np.float32)


def project_masks(pcd, hyps, pivot, H=IMG_H, W=IMG_W):
    """opt_utils.py:431-456: transform, project, truncate, clamp, scatter -> [A,H,W] bool.  fp32 with one rounding per
    operation and sums taken left to right (x' = ((r0*qx + r1*qy) + r2*qz) + pivot + t): where a re-projected point lands
    exactly on a pixel edge (the identity hypothesis puts EVERY point there) the truncation amplifies the last bit, so the
    evaluation order is part of the parity definition."""
    out = np.zeros((len(hyps), H, W), dtype=bool)
    pv = np.asarray(pivot, dtype=np.float32)
    q = (pcd - pv[None, :]).astype(np.float32)
    for a, (R, t) in enumerate(hyps):
        R = R.astype(np.float32)
        t = t.astype(np.float32)
        p = np.stack([((R[i, 0] * q[:, 0] + R[i, 1] * q[:, 1]) + R[i, 2] * q[:, 2]) + pv[i] + t[i] for i in range(3)], 1).astype(np.float32)
        with np.errstate(all="ignore"):
            uv = project2d(p, H, W)
            big = np.float32(2147483520.0)
            uv = np.where(np.isnan(uv), np.float32(0), np.clip(uv, -big, big))
            col = np.clip(uv[:, 0].astype(np.int64), 0, W - 1)
            row = np.clip(uv[:, 1].astype(np.int64), 0, H - 1)
        out[a, row, col] = True
    return out


def mask_ious(target, proj):
    """opt_utils.py:470-475."""
    t = np.asarray(target) > 0.5
    inter = (t[None] & proj).sum((1, 2)).astype(np.float32)
    union = (t[None] | proj).sum((1, 2)).astype(np.float32)
    with np.errstate(all="ignore"):
        return inter / union


ROT_ANGLES = np.arange(-np.pi / 2, np.pi, np.pi / 30).astype(np.float32)        # opt_utils.py:424-426 (45 hypotheses)
ROT_ANGLES_FINAL = np.arange(-np.pi / 2, np.pi / 2, np.pi / 30).astype(np.float32)  # :561-563 (30)
TRANS_STEPS = __import__("torch").arange(-1, 1, 0.1).numpy()                       # the reference's own expression (:723): 20 steps;
#   torch evaluates start + i * float32(0.1) in double, so step 10 is 1.49e-08, not 0


def _sweep(pred, box_id, kind, final=False):
    centers = np.stack([(pred["boxes"][:, 0] + pred["boxes"][:, 2]) / 2, (pred["boxes"][:, 1] + pred["boxes"][:, 3]) / 2], 1)
    if kind == "rot":
        pts = angle_offset_to_axis(pred["rot_axis"], centers)
    else:
        pts = angle_offset_to_axis(np.concatenate([pred["tran_axis"], np.zeros((len(pred["tran_axis"]), 1), np.float32)], 1), centers)
    normal, offset, axis3d, d, pcd = plane_geometry(pred["masks"][box_id], pred["planes"][box_id], pts[box_id])
    if kind == "rot":
        angles = ROT_ANGLES_FINAL if final else ROT_ANGLES
        hyps, pivot = rotation_hypotheses(angles, d, axis3d[0])
    else:
        angles = TRANS_STEPS
        hyps, pivot = translation_hypotheses(angles, d)
    return project_masks(pcd, hyps, pivot), angles, pts[box_id]


def optimize_track(preds: List[dict], plane: dict, kind: str, rng: random.Random):
    """One track of optimize_planes_3dc (kind 'rot', opt_utils.py:386-634) / optimize_planes_3d_trans ('trans', :689-907).
    preds[i] = dict(boxes [n,4], masks [n,H,W], planes [n,3], rot_axis [n,3], tran_axis [n,2]) (numpy)."""
    id_list = list(plane["ids"].keys())
    clusters = []
    for _ in range(5):
        if not id_list:
            break
        sel = rng.choice(id_list)
        proj, angles, _ = _sweep(preds[sel], plane["ids"][sel], kind)
        inl, angs, ious_kept = [], [], []
        for idx in id_list:  # the reference removes from the list it iterates over (:484): the element after a removal is skipped
            ious = mask_ious(preds[idx]["masks"][plane["ids"][idx]], proj)
            if np.nanmax(ious) > 0.5 if np.isfinite(ious).any() else False:
                inl.append(idx)
                id_list.remove(idx)
                angs.append(float(angles[int(np.nanargmax(ious))]))
                ious_kept.append(float(np.nanmax(ious)))
        clusters.append(dict(center_id=sel, inliners=inl, angles=np.asarray(angs, dtype=np.float32), ious=ious_kept))
    rsqs = np.asarray([0.0 if len(c["inliners"]) < 5 else linregress(range(len(c["angles"])), c["angles"]).rvalue ** 2 for c in clusters])
    if rsqs.max() < 0.3:
        plane["has_rot"] = False
        return plane
    plane["has_rot"] = True
    final = clusters[int(rsqs.argmax())]
    sel = final["center_id"]
    box_id = plane["ids"][sel]
    proj, angles, axis_pts = _sweep(preds[sel], box_id, kind, final=True)
    plane["reg_masks"] = {}
    for idx in plane["ids"]:
        ious = mask_ious(preds[idx]["masks"][plane["ids"][idx]], proj)
        plane["reg_masks"][idx] = proj[int(np.nanargmax(ious))]
    plane["std_axis"] = axis_pts if kind == "rot" else preds[sel]["tran_axis"][box_id].copy()
    plane["center_id"] = sel
    return plane
