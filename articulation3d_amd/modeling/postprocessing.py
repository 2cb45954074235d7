"""`detector_postprocess` with the reference's signature (pkg/modeling/postprocessing.py:11-75) and
`paste_masks_in_image` (pkg/layers/mask_ops.py:68-135).  Score filter, clip, empty-box filter and the mask
paste + threshold run in the fused HIP kernel `a3d_paste_lsq` (the plane-offset half is skipped here)."""
from __future__ import annotations

import torch

from .. import ops
from ..structures import Boxes, Instances


def paste_masks_in_image(masks, boxes, image_shape, threshold=0.5, nms=False):
    """masks [N, M, M] probabilities, boxes Boxes/Tensor [N,4] -> bool [N, H, W]."""
    assert masks.shape[-1] == masks.shape[-2], "Only square mask predictions are supported"
    assert not nms, "ROI_MASK_HEAD.NMS is false in every reference config"
    assert threshold >= 0, "the visualisation-only uint8 branch (threshold < 0) is not on the hot path"
    N = len(masks)
    h, w = int(image_shape[0]), int(image_shape[1])
    if not isinstance(boxes, torch.Tensor):
        boxes = boxes.tensor
    if N == 0:
        return torch.zeros((0, h, w), dtype=torch.bool, device=masks.device)
    assert len(boxes) == N, boxes.shape
    dev = masks.device
    b = boxes.reshape(1, N, 4).contiguous().float()
    scores = torch.ones((1, N), device=dev, dtype=torch.float32)
    count = torch.tensor([N], device=dev, dtype=torch.int32)
    off = torch.zeros((2,), device=dev, dtype=torch.int32)
    m, _planes, _area, _keep, _ob = ops.paste_lsq(b, scores, count, off, masks.contiguous().float(), None, None, (h, w),
                                                   post_score_thresh=-1.0, mask_thresh=threshold, clip_boxes=False)
    return m[0].to(torch.bool)


def detector_postprocess(results: Instances, output_height, output_width, mask_threshold=0.5, box_score_threshold=0.7, nms=False):
    scale_x, scale_y = output_width / results.image_size[1], output_height / results.image_size[0]
    fields = results.get_fields()
    selected = fields["scores"] >= box_score_threshold
    results = Instances((output_height, output_width), **{k: v[selected] for k, v in fields.items()})
    if results.has("pred_boxes"):
        output_boxes = results.pred_boxes
    elif results.has("proposal_boxes"):
        output_boxes = results.proposal_boxes
    output_boxes.scale(scale_x, scale_y)
    output_boxes.clip(results.image_size)
    results = results[output_boxes.nonempty()]
    if results.has("pred_masks"):
        results.pred_masks = paste_masks_in_image(results.pred_masks[:, 0, :, :], results.pred_boxes, results.image_size,
                                                  threshold=mask_threshold, nms=nms)
    return results
