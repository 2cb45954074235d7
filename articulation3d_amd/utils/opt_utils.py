"""Host-side consumer of the per-frame detections: the IoU tracker that opens the temporal optimiser.

`track_planes` restates pkg/utils/opt_utils.py:1156-1208 (greedy association of each detection with the first live
track of its articulation type whose last box overlaps by IoU > 0.5, tracks die after a gap of more than 5 frames,
tracks shorter than 10 frames are dropped).  It stays on the host by design (BASELINE north star); it consumes the
list[Instances] that `pipeline.detect_clip` rebuilds from the all-gathered detection records.
The clustering passes `optimize_planes_3dc / _3d_trans` (opt_utils.py:382-959) are SURVEY.md 8f-3 ("next").
"""
from __future__ import annotations

from typing import Dict, List

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
