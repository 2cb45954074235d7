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
from .parallel import gather_records, shard_range
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
                resize_to=(480, 640)) -> List[Instances]:
    """frames: uint8 [F,H,W,3] BGR (numpy or tensor).  Returns the detections of ALL F frames in temporal order on every
    rank (single process: plain batching).
    source_rgb=True: frames are what the video reader hands over -- RGB, any size; the reference's host-side
    `cv2.resize(im, (640, 480))` and BGR flip (tools/inference.py:216-218) then run on the device, fused with the
    normalisation in front of the stem (a3d_preprocess_resize_u8)."""
    if isinstance(frames, np.ndarray):
        frames = torch.from_numpy(frames)
    F_ = frames.shape[0]
    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
    rank = torch.distributed.get_rank() if dist_on else 0
    world = torch.distributed.get_world_size() if dist_on else 1
    per = (F_ + world - 1) // world
    lo, hi = shard_range(F_, rank, world)
    hw = tuple(resize_to) if source_rgb else tuple(frames.shape[1:3])
    dev = model.device
    R = model.roi_heads.box_predictor.test_topk_per_image
    rec_f = ops.record_floats(28)
    # fixed-size block per rank so the gather is one collective (uneven shards are zero-padded to `per` frames)
    block_rec = torch.zeros((per, R, rec_f), device=dev)
    block_cnt = torch.zeros((per,), device=dev, dtype=torch.int32)
    for s in range(lo, hi, batch):
        e = min(s + batch, hi)
        out = model.inference_batched(frames[s:e].to(dev, non_blocking=True).contiguous(), source_rgb=source_rgb, resize_to=resize_to)
        block_rec[s - lo:e - lo] = out.records
        block_cnt[s - lo:e - lo] = out.rec_count
    all_rec, all_cnt = gather_records(block_rec, block_cnt, rows=per)
    return instances_from_records(all_rec, all_cnt, [(r * per + i) for r in range(world) for i in range(shard_range(F_, r, world)[1] - shard_range(F_, r, world)[0])],
                                  hw, conf_threshold, with_masks)


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
