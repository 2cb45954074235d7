#!/usr/bin/env python3
"""Dev tool (GPU box): the three ROI poolers on the detector's own proposals / detections of a synthetic clip: time per launch,
algorithmic bytes, achieved GB/s, and bit equality of the batched-load form with the serialized one (ops.ROI_SERIAL = True).
    python tools/roi_bench.py [--frames 64]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    args = ap.parse_args()
    from bench import build_detector
    from articulation3d_amd import ops
    from articulation3d_amd.utils.synthetic import synthetic_frames

    model, _ = build_detector(0.0, "cuda:0")
    fr = torch.from_numpy(synthetic_frames(args.frames, 2020)).cuda()
    x4 = ops.preprocess_u8hwc(fr, model.pixel_mean, model.pixel_std)
    feats = model.backbone.forward_nhwc(x4)
    pb, _pl, _lv, _pos, pc = model.proposal_generator.forward_batched(feats, (480, 640))
    rh = model.roi_heads
    lv = [feats[f] for f in rh.box_in_features]
    scales = rh.box_pooler.scales
    det = rh.box_batched(feats, pb, pc, (480, 640))
    cases = [("box 7x7 aligned", pb, pc, 7, 0, True), ("mask 14x14 ratio 2", det.boxes, det.count, 14, 2, False),
             ("plane/axis 14x14 adaptive", det.boxes, det.count, 14, 0, False)]
    for name, boxes, count, P, ratio, aligned in cases:
        n = int(count.sum())
        f = lambda: ops.roi_align_fpn(lv, scales, boxes, count, P, ratio, aligned)
        ops.ROI_ROLLING = False  # (the three bin-by-bin forms first: they agree bit for bit)
        ops.ROI_SERIAL = True
        y_s = f()
        t_s = timeit(f)
        ops.ROI_SERIAL = False
        ops.ROI_SPATIAL_ORDER = False
        y_u = f()
        t_u = timeit(f)
        ops.ROI_SPATIAL_ORDER = True
        y_b = f()
        t_b = timeit(f)
        y_r, t_r = None, float("nan")
        if P == 7:
            ops.ROI_ROLLING = True
            y_r = f()
            t_r = timeit(f)
        ops.ROI_ROLLING = True
        R = boxes.shape[1]
        live = torch.cat([torch.arange(int(c)) + b * R for b, c in enumerate(count.tolist())]).cuda()
        same = bool(torch.equal(y_s[live], y_b[live]) and torch.equal(y_u[live], y_b[live]))
        wbytes = n * P * P * 256 * 4
        print(f"{name:28s} rois {n:6d}  serialized {t_s:7.3f} ms  batched {t_u:7.3f} ms  + spatial order {t_b:7.3f} ms  output {wbytes / 1e9:5.2f} GB -> {wbytes / t_b / 1e6:7.1f} GB/s written"
              f"  bit-identical={same}" + ("" if y_r is None else f"  | rolling window {t_r:7.3f} ms, max |diff| / max {float((y_r[live] - y_b[live]).abs().max() / y_b[live].abs().max()):.2e}"), flush=True)


if __name__ == "__main__":
    main()
