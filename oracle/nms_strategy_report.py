"""Does torchvision's coordinate-offset `batched_nms` keep a different set than per-category NMS on the committed frames?

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  SURVEY.md section 7 defines parity as one NMS per category on the boxes as
they are ("plain"); torchvision >= 0.9 -- what the reference runs through detectron2's `batched_nms` at
pkg/modeling/meta_arch/planercnn.py:168 (RPN) and pkg/modeling/roi_heads/roi_heads.py:206 (box stage) -- shifts every category by
`idx * (boxes.max() + 1)` and runs ONE nms while a call holds at most 5000 boxes on a GPU / 1000 on the CPU
(`planercnn_oracle.batched_nms`).  The two agree unless an IoU lies within fp32 rounding (~1e-6) of the threshold.  This script
evaluates every committed frame -- the end-to-end seeds of tests/golden/e2e_frames.json, the stage-test frames (seed 2020) and
the smoke frame (seed 3000) -- under all four strategies, reusing one backbone evaluation per frame, and records every
difference in the kept proposals and in the final detections.

    python -m oracle.nms_strategy_report            # writes tests/golden/nms_strategy_report.json
"""
from __future__ import annotations

import json
import os
import sys

import torch

from . import planercnn_oracle as O
from . import matching as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRATEGIES = ("plain", "offset", "tv-gpu", "tv-cpu")


def frame_report(seed: int, P, thresholds=(0.5, 0.0), index: int = 0):
    fr = O.synthetic_frames(index + 1, seed=seed)[index:index + 1]
    imgs = O.frames_to_chw(fr)
    x, sizes = O.preprocess(imgs, O.OracleCfg())
    feats = O.backbone(x, P)
    rep = dict(seed=seed, index=index)
    for t in thresholds:
        base = None
        for sname in STRATEGIES:
            cfg = O.OracleCfg(score_thresh=t, nms_strategy=sname)
            out, aux = O.detect(imgs, P, cfg, return_aux=True, features=feats)
            props = aux["proposals"][0][0]
            if base is None:
                base = (out[0], props)
                continue
            same_props = props.shape == base[1].shape and bool(torch.equal(props, base[1]))
            same_det = M.same_discrete_result(base[0], out[0], box_px=0.0) and bool(torch.equal(base[0]["scores"], out[0]["scores"]))
            rep[f"t{t}_{sname}"] = dict(proposals_identical=same_props, detections_identical=same_det,
                                        proposals=int(props.shape[0]), detections=int(len(out[0]["scores"])))
    return rep


def main():
    torch.set_grad_enabled(False)
    P = O.init_params(2020)
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "e2e_frames.json")))
    jobs = [(f["seed"], 0) for f in doc["frames"]] + [(2020, 0), (2020, 1), (3000, 0)]
    frames = []
    for seed, idx in jobs:
        r = frame_report(seed, P, index=idx)
        frames.append(r)
        print(json.dumps(r), flush=True)
    flips = [(r["seed"], r["index"], k) for r in frames for k, v in r.items() if isinstance(v, dict) and not (v["proposals_identical"] and v["detections_identical"])]
    out = dict(strategies=STRATEGIES, frames=frames, differing=[list(f) for f in flips],
               summary=f"{len(frames)} committed frames x 2 thresholds x 3 alternative strategies: {len(flips)} differ from per-category NMS")
    with open(os.path.join(ROOT, "tests", "golden", "nms_strategy_report.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(out["summary"])


if __name__ == "__main__":
    sys.exit(main())
