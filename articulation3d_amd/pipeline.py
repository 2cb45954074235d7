"""Batched, frame-sharded replacement of the reference's per-frame hot loop (tools/inference.py:215-228).

    preds = detect_clip(model, frames)          # list[Instances], one per frame, the `create_instances` format
    planes = track_planes(preds)                # host-side temporal stage

Frames go through `PlaneRCNN.inference_batched` in batches (uint8, device-resident), each rank of a
torch.distributed job taking one contiguous block of the clip (parallel.shard_range); the packed detection records
are all-gathered (parallel.gather_records) and every rank -- rank 0 in practice -- rebuilds the per-frame
`Instances` the optimiser consumes: scores, boxes, classes, plane normal*offset, rotation / translation axes and the
480x640 masks, which the receiving side re-pastes from the 28x28 soft masks with the same kernel the sender would have
used (bit-identical).  This fuses `process` + `create_instances` (pkg/utils/arti_vis.py:63-194) on the device and drops
their RLE encode -> decode -> decode round trips.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from . import ops
from .modeling.postprocessing import paste_masks_in_image
from .parallel import gather_records, gather_records_async, shard_range
from .structures import Boxes, Instances


def instances_from_record(rec: torch.Tensor, image_size, conf_threshold: float = 0.7, with_masks: bool = True) -> Instances:
    """rec: [n, 798] live records of ONE frame (device or cpu) -> the Instances of create_instances (arti_vis.py:152-194):
    strict `score > conf_threshold`, numpy scores / classes, cpu tensors for the rest."""
    score = rec[:, 4].double()
    chosen = (score > conf_threshold).nonzero().squeeze(1)
    rec = rec[chosen]
    ret = Instances(tuple(image_size))
    ret.scores = score[chosen].cpu().numpy()
    ret.pred_boxes = Boxes(rec[:, 0:4].cpu())
    ret.pred_classes = rec[:, 5].cpu().numpy().astype(np.int64)
    ret.pred_planes = rec[:, 6:9].cpu()
    ret.pred_rot_axis = rec[:, 9:12].cpu()
    ret.pred_tran_axis = rec[:, 12:14].cpu()
    if with_masks:
        n = rec.shape[0]
        ms = int(round((rec.shape[1] - 14) ** 0.5))
        if n and rec.is_cuda:
            masks = paste_masks_in_image(rec[:, 14:].reshape(n, ms, ms).contiguous(), rec[:, 0:4].contiguous(), image_size)
            ret.pred_masks = masks.float().cpu()
        else:
            assert n == 0, "re-pasting masks needs the records on the device"
            ret.pred_masks = torch.zeros((0,) + tuple(image_size))
    return ret


@torch.no_grad()
def detect_clip(model, frames, batch: int = 32, conf_threshold: float = 0.7, with_masks: bool = True, source_rgb: bool = False,
                resize_to=(480, 640), num_frames: Optional[int] = None) -> List[Instances]:
    """frames: uint8 [F,H,W,3] BGR (numpy or tensor).  Returns the detections of ALL F frames in temporal order on every
    rank (single process: plain batching).
    source_rgb=True: frames are what the video reader hands over -- RGB, any size; the reference's host-side
    `cv2.resize(im, (640, 480))` and BGR flip (tools/inference.py:216-218) then run on the device, fused with the
    normalisation in front of the stem (a3d_preprocess_resize_u8).
    num_frames: `frames` holds ONLY this rank's block `shard_range(num_frames, rank, world)` of a clip of that many frames (each
    rank of tools/inference.py reads its own shard instead of the whole clip)."""
    if isinstance(frames, np.ndarray):
        frames = torch.from_numpy(frames)
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    rank = torch.distributed.get_rank() if dist_on else 0
    world = torch.distributed.get_world_size() if dist_on else 1
    F_ = int(num_frames) if num_frames is not None else frames.shape[0]
    per = (F_ + world - 1) // world
    lo, hi = shard_range(F_, rank, world)
    base = lo if num_frames is not None else 0  # index of frames[0] in the clip
    assert frames.shape[0] >= hi - base, (frames.shape[0], lo, hi)
    hw = tuple(resize_to) if source_rgb else tuple(frames.shape[1:3])
    dev = model.device
    R = model.roi_heads.box_predictor.test_topk_per_image
    rec_f = ops.record_floats(28)
    # fixed-size block per rank (uneven shards are padded with empty frames: count 0); only the LIVE records travel
    block_rec = torch.zeros((per, R, rec_f), device=dev)
    block_cnt = torch.zeros((per,), device=dev, dtype=torch.int32)
    for s in range(lo, hi, batch):
        e = min(s + batch, hi)
        out = model.inference_batched(frames[s - base:e - base].to(dev, non_blocking=True).contiguous(), source_rgb=source_rgb, resize_to=resize_to)
        block_rec[s - lo:e - lo] = out.records
        block_cnt[s - lo:e - lo] = out.rec_count
    rows, all_cnt = gather_records_async(block_rec, block_cnt, rows=per).wait_compact()
    slots = [(r * per + i) for r in range(world) for i in range(shard_range(F_, r, world)[1] - shard_range(F_, r, world)[0])]
    return instances_from_compact(rows, all_cnt, slots, hw, conf_threshold, with_masks)


def instances_from_compact(rows: torch.Tensor, all_cnt: torch.Tensor, slots, image_size, conf_threshold: float = 0.7,
                           with_masks: bool = True) -> List[Instances]:
    """`instances_from_records` on the compact form of the gathered records (parallel.GatherHandle.wait_compact): rows [N, 798] =
    the live records of every slot in slot order, all_cnt [slots] their per-slot counts; `slots` = the slot of every frame in
    temporal order (increasing: rank blocks are contiguous)."""
    counts = all_cnt.tolist()
    start = [0]
    for c in counts:
        start.append(start[-1] + int(c))
    keep_score = (rows[:, 4].double() > conf_threshold) if rows.shape[0] else torch.zeros(0, dtype=torch.bool, device=rows.device)
    frame_of = torch.full((rows.shape[0],), -1, dtype=torch.int64)
    for f, s in enumerate(slots):
        frame_of[start[s]:start[s + 1]] = f
    frame_of = frame_of.to(rows.device)
    sel = keep_score & (frame_of >= 0)
    kept = rows[sel]
    per_frame = torch.bincount(frame_of[sel], minlength=len(slots)).tolist() if kept.shape[0] else [0] * len(slots)
    return _instances_of_rows(kept, per_frame, image_size, with_masks)


def instances_from_records(all_rec: torch.Tensor, all_cnt: torch.Tensor, slots, image_size, conf_threshold: float = 0.7,
                           with_masks: bool = True) -> List[Instances]:
    """The per-frame `create_instances` records of a whole clip from its gathered detection records, with ONE re-paste launch
    and ONE device -> host copy for all frames (the per-frame form issues a launch, a 1.2 MB-per-detection float copy and a
    synchronisation for every frame).  `slots`: index into all_rec / all_cnt of every frame, in temporal order."""
    counts = all_cnt.tolist()
    R = all_rec.shape[1]
    live = torch.arange(R, device=all_rec.device)[None, :] < all_cnt[:, None]
    keep = live & (all_rec[:, :, 4].double() > conf_threshold)          # strict `score > threshold` (arti_vis.py:156)
    sel = keep[torch.as_tensor(slots, device=all_rec.device)]           # [F, R] in temporal order
    rows = all_rec[torch.as_tensor(slots, device=all_rec.device)][sel]  # [N, 798] kept records of the clip, frame-major
    per_frame = sel.sum(1).tolist()
    return _instances_of_rows(rows, per_frame, image_size, with_masks)


def _instances_of_rows(rows: torch.Tensor, per_frame, image_size, with_masks: bool) -> List[Instances]:
    """rows [N, 798]: the kept records of a clip, frame-major; per_frame: how many belong to each frame."""
    from .structures import to_host

    rec_h = to_host(rows)
    masks_h = None
    if with_masks and rows.shape[0]:
        assert rows.is_cuda, "re-pasting masks needs the records on the device"
        n, ms = rows.shape[0], int(round((rows.shape[1] - 14) ** 0.5))
        masks = paste_masks_in_image(rows[:, 14:].reshape(n, ms, ms).contiguous(), rows[:, 0:4].contiguous(), image_size)
        masks_h = to_host(masks.to(torch.uint8))  # uint8 over PCIe (4x fewer bytes than float), float on the host as create_instances returns
    preds, o = [], 0
    for n in per_frame:
        rec = rec_h[o:o + n]
        ret = Instances(tuple(image_size))
        ret.scores = rec[:, 4].double().numpy()
        ret.pred_boxes = Boxes(rec[:, 0:4].clone())
        ret.pred_classes = rec[:, 5].numpy().astype(np.int64)
        ret.pred_planes = rec[:, 6:9].clone()
        ret.pred_rot_axis = rec[:, 9:12].clone()
        ret.pred_tran_axis = rec[:, 12:14].clone()
        if with_masks:
            ret.pred_masks = masks_h[o:o + n].float() if masks_h is not None and n else torch.zeros((0,) + tuple(image_size))
        preds.append(ret)
        o += n
    return preds
