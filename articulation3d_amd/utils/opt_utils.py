"""Host-side consumer of the per-frame detections: the IoU tracker that opens the temporal optimiser.

`track_planes` restates pkg/utils/opt_utils.py:1156-1208 (greedy association of each detection with the first live
track of its articulation type whose last box overlaps by IoU > 0.5, tracks die after a gap of more than 5 frames,
tracks shorter than 10 frames are dropped).  It stays on the host by design (BASELINE north star); it consumes the
list[Instances] that `pipeline.detect_clip` rebuilds from the all-gathered detection records.
`optimize_planes` (below) restates the '3dc' method (opt_utils.py:382-975) with its hypothesis sweeps on the GPU
(SURVEY.md 8f-3).
"""
from __future__ import annotations

import random as _random
from typing import Dict, List

import numpy as np
import torch
from scipy.stats import linregress

from ..structures import Instances, pairwise_iou

MAX_GAP_FRAMES = 5      # opt_utils.py:1177
MATCH_IOU = 0.5         # opt_utils.py:1181
MIN_TRACK_FRAMES = 10   # opt_utils.py:1203


def track_planes(preds: List[Instances]) -> Dict[str, list]:
    planes = {"rot": [], "trans": []}
    for idx, inst in enumerate(preds):
        boxes, classes = inst.pred_boxes, inst.pred_classes
        for box_id in range(len(boxes)):
            current = boxes[box_id]
            cat = "trans" if int(classes[box_id]) == 1 else "rot"
            matched = False
            for track in planes[cat]:
                if idx - track["latest_frame"] > MAX_GAP_FRAMES:
                    continue
                if pairwise_iou(current, track["bbox"]).item() > MATCH_IOU:
                    track["ids"][idx] = box_id
                    track["bbox"] = current
                    track["latest_frame"] = idx
                    matched = True
                    break
            if not matched:
                planes[cat].append({"bbox": current, "ids": {idx: box_id}, "latest_frame": idx})
    return {cat: [t for t in tracks if len(t["ids"]) >= MIN_TRACK_FRAMES] for cat, tracks in planes.items()}


# ======================================================================================================
# optimize_planes(preds, planes, '3dc')  (pkg/utils/opt_utils.py:962-975): the temporal optimiser.
# Control flow (random cluster centres, inlier bookkeeping, linear regression of the per-frame best hypothesis, score
# re-weighting) stays on the host as in the reference; the two sweeps it spends its time in -- "project this
# detection's mask under 45 rotations / 20 translations of its plane about its axis" and "IoU of every projection with
# every tracked mask" -- run as two GPU launches on bit-packed masks (include/a3d.h: a3d_project_hypotheses,
# a3d_mask_iou_matrix) instead of a Python loop with one scatter and two full-image reductions per (hypothesis, frame).
# ======================================================================================================
FOCAL = 517.97  # pkg/utils/vis.py:62,86 (the optimiser's intrinsics; the detector's ray table uses 571.62)


def _get_pcd(verts, normal, offset, h=480, w=640):
    """get_pcd (vis.py:86-102), float64 on the host: used for the two axis end points only."""
    K_inv = np.linalg.inv(np.array([[FOCAL, 0, w / 2], [0, FOCAL, h / 2], [0, 0, 1]]))
    ray = K_inv @ np.hstack((np.asarray(verts, dtype=np.float64), np.ones((len(verts), 1)))).T
    depth = float(offset) / (np.asarray(normal, dtype=np.float64) @ ray)
    return depth.reshape(-1, 1) * ray.T


def get_boundary_point(y, x, angle, H, W):
    """planercnn_transforms.py:131-176."""
    if float(angle) == -np.pi / 2:  # (the reference compares with the float64 constant: only ITS OWN sentinel for sin == 0 matches, never a float32 arctan)
        return (x, 0), (x, H - 1)
    if angle == 0.0:
        return (0, y), (W - 1, y)
    k = np.tan(angle)
    cands = []
    if 0 <= y - k * x < H:
        cands.append((0, int(y - k * x)))
    if 0 <= k * (W - 1) + y - k * x < H:
        cands.append((W - 1, int(k * (W - 1) + y - k * x)))
    if 0 <= x - y / k < W:
        cands.append((int(x - y / k), 0))
    if 0 <= x - y / k + (H - 1) / k < W:
        cands.append((int(x - y / k + (H - 1) / k), H - 1))
    p1 = p2 = None
    for c in cands:  # first two DISTINCT candidates, in the reference's left / right / top / bottom order
        if p1 is None:
            p1 = c
        elif p2 is None and c != p1:
            p2 = c
    return p1, (p2 if p2 is not None else p1)


def angle_offset_to_axis(angle_offsets: torch.Tensor, centers: torch.Tensor, H=480, W=640) -> torch.Tensor:
    """planercnn_transforms.py:101-129: (sin, cos, offset/100) per box + box centre -> integer axis end points [n,4]."""
    rtn = []
    for ao, c in zip(angle_offsets.detach().cpu().numpy().astype(np.float32), centers.detach().cpu().numpy().astype(np.float32)):
        sin, cos, p = ao[0], ao[1], np.float32(ao[2] * np.float32(100))
        angle = -np.pi / 2 if sin == 0 else np.float32(-np.arctan(cos / sin))
        x, y = np.float32(p * cos + c[0]), np.float32(p * sin + c[1])
        p1, p2 = get_boundary_point(y, x, angle, H, W)
        rtn.append([0, 0, 1, 1] if p1 is None else [p1[0], p1[1], p2[0], p2[1]])
    return torch.tensor(np.asarray(rtn, dtype=np.float64)).long()


def axis_to_angle_offset(axis, center: torch.Tensor) -> torch.Tensor:
    """planercnn_transforms.py:31-68 (mine=False)."""
    a = torch.FloatTensor(axis) - torch.cat((center, center), dim=1)
    x1, y1, x2, y2 = a[:, :1], a[:, 1:2], a[:, 2:3], a[:, 3:4]
    A, B, Cc = y1 - y2, x2 - x1, x1 * y2 - x2 * y1
    lll = torch.sqrt(A * A + B * B)
    offset = torch.abs(Cc) / lll / 100
    return torch.cat((-B * torch.sign(Cc) / lll, -A * torch.sign(Cc) / lll, offset, torch.ones_like(offset)), dim=1)


def _axis_angle_to_matrix(axis_angle: np.ndarray) -> np.ndarray:
    """pytorch3d.transforms.axis_angle_to_matrix (axis-angle -> quaternion -> matrix), float64."""
    ang = np.linalg.norm(axis_angle, axis=-1, keepdims=True)
    small = np.abs(ang) < 1e-6
    s = np.where(small, 0.5 - ang * ang / 48, np.sin(0.5 * ang) / np.where(small, 1.0, ang))
    q = np.concatenate([np.cos(0.5 * ang), axis_angle * s], -1)
    r, i, j, k = (q[..., n] for n in range(4))
    ts = 2.0 / (q * q).sum(-1)
    m = np.stack((1 - ts * (j * j + k * k), ts * (i * j - k * r), ts * (i * k + j * r), ts * (i * j + k * r), 1 - ts * (i * i + k * k),
                  ts * (j * k - i * r), ts * (i * k - j * r), ts * (j * k + i * r), 1 - ts * (i * i + j * j)), -1)
    return m.reshape(q.shape[:-1] + (3, 3))


ROT_ANGLES = torch.FloatTensor(np.arange(-np.pi / 2, np.pi, np.pi / 30))            # opt_utils.py:424-426
ROT_ANGLES_FINAL = torch.FloatTensor(np.arange(-np.pi / 2, np.pi / 2, np.pi / 30))  # :561-563
TRANS_STEPS = torch.arange(-1, 1, 0.1)                                              # :723


class _MaskBank:
    """The tracked detections' masks of one clip, bit-packed on the device once (9 600 words per 480x640 mask)."""

    def __init__(self, preds, device, wanted=None):
        """`wanted`: set of (frame index, box id) to pack -- the detections that belong to a track; None packs everything.
        (A long clip keeps one 307 KB mask per kept detection on the device otherwise: only tracked ones are ever read.)"""
        from .. import opt_ops

        self.dev, self.index, chunks = device, {}, []
        n = 0
        for idx, inst in enumerate(preds):
            m = inst.pred_masks
            if m is None or len(m) == 0:
                continue
            keep = [b for b in range(len(m)) if wanted is None or (idx, b) in wanted]
            if not keep:
                continue
            for j, b in enumerate(keep):
                self.index[(idx, b)] = n + j
            n += len(keep)
            chunks.append((m[keep] > 0.5).to(torch.uint8))
        self.H, self.W = (chunks[0].shape[1], chunks[0].shape[2]) if chunks else (480, 640)
        self.u8 = torch.cat(chunks).to(device) if chunks else torch.zeros((0, self.H, self.W), dtype=torch.uint8, device=device)
        self.bits = opt_ops.pack_masks(self.u8) if n else None

    def rows(self, keys):
        return torch.tensor([self.index[k] for k in keys], device=self.dev, dtype=torch.long)


def sweep_hypotheses(bank: _MaskBank, p_instance, box_id: int, frame_idx: int, kind: str, final: bool = False):
    """One sweep of the reference (opt_utils.py:400-456 / 700-748): -> (projected bit masks [A,words] on the device,
    the hypothesis parameters (angles / steps), the integer axis end points of the centre detection)."""
    from .. import opt_ops

    plane = p_instance.pred_planes[box_id].clone().float()
    plane = torch.stack((plane[0], -plane[2], plane[1]))  # (a, b, c) -> (a, -c, b)
    offset = torch.norm(plane, p=2)
    normal = torch.nn.functional.normalize(plane[None], p=2)[0]
    centers = p_instance.pred_boxes.get_centers()
    if kind == "rot":
        pts = angle_offset_to_axis(p_instance.pred_rot_axis, centers)
    else:
        at = p_instance.pred_tran_axis
        pts = angle_offset_to_axis(torch.cat((at, torch.zeros(len(at), 1)), 1), centers)
    axis3d = _get_pcd(pts[box_id].reshape(-1, 2).numpy(), normal.numpy(), offset.item())
    d = axis3d[1] - axis3d[0]
    d = d / np.linalg.norm(d)
    if kind == "rot":
        params = ROT_ANGLES_FINAL if final else ROT_ANGLES
        R = _axis_angle_to_matrix(params.double().numpy()[:, None] * d[None, :]).astype(np.float32)
        # pytorch3d's Rotate multiplies row vectors (points @ R): the effective column-vector rotation is R^T
        xf = np.concatenate([np.transpose(R, (0, 2, 1)).reshape(-1, 9), np.zeros((len(R), 3), np.float32)], 1)
        pivot = axis3d[0].astype(np.float32)
    else:
        params = TRANS_STEPS
        t = (params.double().numpy()[:, None] * d[None, :]).astype(np.float32)
        xf = np.concatenate([np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (len(t), 1)), t], 1)
        pivot = np.zeros(3, np.float32)
    src = bank.u8[bank.index[(frame_idx, box_id)]]
    proj = opt_ops.project_hypotheses(src, normal.tolist(), offset.item(), pivot.tolist(), torch.from_numpy(np.ascontiguousarray(xf)).to(bank.dev),
                                      focal=FOCAL, cx=bank.W / 2, cy=bank.H / 2)
    return proj, params, pts[box_id]


def _optimize_track(preds, plane, kind, bank: _MaskBank):
    """One tracked plane of optimize_planes_3dc (kind 'rot', :386-634) / optimize_planes_3d_trans ('trans', :689-907)."""
    from .. import opt_ops

    id_list = list(plane["ids"].keys())
    clusters = []
    for _ in range(5):
        if len(id_list) == 0:
            break
        select_idx = _random.choice(id_list)  # the reference seeds `random` once per run (tools/inference.py:172)
        proj, params, _ = sweep_hypotheses(bank, preds[select_idx], plane["ids"][select_idx], select_idx, kind)
        cand = list(id_list)
        ious = opt_ops.mask_iou_matrix(bank.bits[bank.rows([(i, plane["ids"][i]) for i in cand])], proj, bank.H, bank.W).cpu()
        row_of = {i: r for r, i in enumerate(cand)}
        inl, angs, kept = [], [], []
        for idx in id_list:  # (removing from the list being iterated skips the element after each removal, as in the reference)
            row = ious[row_of[idx]]
            if row.max() > 0.5:
                inl.append(idx)
                id_list.remove(idx)
                angs.append(params[row.argmax()])
                kept.append(row.max().item())
        clusters.append({"center_id": select_idx, "inliners": inl, "angles": torch.FloatTensor([float(a) for a in angs]), "ious": kept})
    rsqs = np.array([0.0 if len(c["inliners"]) < 5 else linregress(range(c["angles"].shape[0]), c["angles"]).rvalue ** 2 for c in clusters])
    if rsqs.max() < 0.3:
        plane["has_rot"] = False
        return
    plane["has_rot"] = True
    final = clusters[rsqs.argmax()]
    select_idx = final["center_id"]
    box_id = plane["ids"][select_idx]
    proj, _params, axis_pts = sweep_hypotheses(bank, preds[select_idx], box_id, select_idx, kind, final=True)
    keys = list(plane["ids"].keys())
    ious = opt_ops.mask_iou_matrix(bank.bits[bank.rows([(i, plane["ids"][i]) for i in keys])], proj, bank.H, bank.W)
    best = ious.argmax(1)
    reg = opt_ops.unpack_masks(proj[best].contiguous(), bank.H, bank.W).cpu()
    plane["reg_masks"] = {idx: reg[r] for r, idx in enumerate(keys)}
    plane["std_axis"] = axis_pts if kind == "rot" else preds[select_idx].pred_tran_axis[box_id]
    plane["center_id"] = select_idx


def _reweight(preds, planes, kind):
    """Write the consensus axis back and down-weight detections without a consistent motion (:621-683, :909-959)."""
    keep_class = 1 if kind == "rot" else 0  # the other articulation type is never filtered by this pass
    out = []
    for idx, inst in enumerate(preds):
        n = len(inst.pred_boxes)
        chosen = [int(inst.pred_classes[i]) == keep_class for i in range(n)]
        if kind == "rot":
            inst.pred_rot_axis = inst.pred_rot_axis.clone()
            inst.pred_planes = inst.pred_planes.clone()
        for plane in planes:
            if idx not in plane["ids"]:
                continue
            box_id = plane["ids"][idx]
            if not plane["has_rot"]:
                chosen[box_id] = False
                continue
            chosen[box_id] = True
            if kind == "rot":
                c = inst.pred_boxes.get_centers()[box_id:box_id + 1]
                inst.pred_rot_axis[box_id] = axis_to_angle_offset(plane["std_axis"].unsqueeze(0).numpy().tolist(), c)[0, :3]
            else:
                inst.pred_tran_axis[box_id] = plane["std_axis"]
        new = Instances(inst.image_size)
        scores = np.copy(inst.scores)
        scores[np.logical_not(np.array(chosen, dtype=bool))] *= 0.6
        new.scores = scores
        for f in ("pred_boxes", "pred_planes", "pred_rot_axis", "pred_tran_axis", "pred_masks", "pred_classes"):
            setattr(new, f, getattr(inst, f))
        out.append(new)
    return out


def optimize_planes(preds: List[Instances], planes: Dict[str, list], method: str = "3dc", frames=None, device="cuda"):
    """pkg/utils/opt_utils.py:962-975, method '3dc': translation tracks first, then rotation tracks."""
    if method != "3dc":
        raise NotImplementedError("only the '3dc' method the reference's tools call (tools/inference.py:250) is provided")
    wanted = {(idx, b) for kind in ("trans", "rot") for plane in planes[kind] for idx, b in plane["ids"].items()}
    bank = _MaskBank(preds, device, wanted)  # only tracked detections are ever projected or compared
    for plane in planes["trans"]:
        _optimize_track(preds, plane, "trans", bank)
    preds = _reweight(preds, planes["trans"], "trans")
    for plane in planes["rot"]:
        _optimize_track(preds, plane, "rot", bank)
    return _reweight(preds, planes["rot"], "rot")
