"""Runs the same small batch through inference_batched repeatedly and reports any run whose outputs differ from the first one
(bitwise): a probe for races between the concurrent branches of 1-2 frame batches.  usage: determinism_probe.py [B] [runs] [thresh]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_detector  # noqa: E402
from articulation3d_amd.utils.synthetic import synthetic_frames  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 200
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
model, cfg = build_detector(thr, "cuda:0")
frames = torch.from_numpy(synthetic_frames(B)).cuda()
fields = ("depth", "records", "rec_count", "planes", "boxes", "keep")


def snap(o):
    d = {k: getattr(o, k).clone() for k in fields}
    d["det_boxes"], d["det_scores"], d["det_count"] = o.det.boxes.clone(), o.det.scores.clone(), o.det.count.clone()
    for k in ("mask_prob", "pred_plane", "pred_rot_axis", "pred_tran_axis"):
        v = getattr(o.det, k)
        if v is not None:
            d[k] = v.clone()
    d["prop_boxes"] = o.proposals[0].clone()
    return d


ref = snap(model.inference_batched(frames, want_masks=False))
torch.cuda.synchronize()
bad = 0
for i in range(runs):
    cur = snap(model.inference_batched(frames, want_masks=False))
    torch.cuda.synchronize()
    diff = [k for k in ref if ref[k].shape != cur[k].shape or not torch.equal(ref[k], cur[k])]
    if diff:
        bad += 1
        if bad <= 5:
            det = {k: (float((ref[k].float() - cur[k].float()).abs().max()) if ref[k].shape == cur[k].shape else "shape") for k in diff}
            print(f"run {i}: differs in {det}", flush=True)
print(f"B={B} runs={runs}: {bad} runs differ from the first", flush=True)
