#!/usr/bin/env python3
"""Seed search for the end-to-end "matched detections" test (hard parts: "end-to-end fixtures chosen
(by seed search in the oracle) so that the minimum margin to any threshold/tie exceeds the observed GEMM error; report
margin alongside parity").

TEST INFRASTRUCTURE ONLY.  Runs the CPU oracle alone (no GPU, no product code):

  for every candidate seed s:  frame = synthetic_frames(1, seed=s)
    1. the oracle's detections at every operating point (SCORE_THRESH_TEST 0.5 and 0.0);
    2. STABILITY under the rounding of the fp32 evaluation order: the backbone is evaluated three more times on the CPU --
       (B) oneDNN disabled (ATen's native convolution: another summation order), (C) in float64 and rounded to fp32,
       (D) in channels-last memory format (other oneDNN kernels) -- and everything behind it is re-run on those feature
       maps.  These variants differ from the default evaluation (A) by 3e-5 .. 9e-5 of the level maximum, the same size as
       the HIP path's deviation from (A) measured on the MI355X (4e-5 .. 8.5e-5).  A seed is STABLE when all variants yield
       the same detections at both operating points: same count, and rank for rank (ranks may be exchanged only inside a
       group of scores tied to 2e-4) the same class and the same box within 0.05 px;
       (independent Gaussian noise of the same rms on the feature maps was tried first and is far harsher than any real
       evaluation order: it flips 100 % of the frames, while the MI355X run agrees with (A) on 40/40 frames at 0.5 and 35/40
       at 0.0 -- rounding noise that has been propagated through the network is not white);
    3. MARGINS: distance of every discrete decision of run (A) to its threshold / tie
       (oracle/matching.py:decision_margins), recorded next to the seed.

Writes tests/golden/e2e_frames.json: the first N stable seeds with their margins.  The GPU test
(tests/test_gpu_e2e.py) regenerates the frames from the seeds; no reference file is involved (the frames are synthetic,
SURVEY.md 8d).

    python -m oracle.seed_search [--first 3000] [--candidates 40] [--want 8]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import matching as M  # noqa: E402
from oracle import planercnn_oracle as O  # noqa: E402

THRESHOLDS = (0.5, 0.0)


def backbone_variants(x, P, P64):
    """(name, feature maps) of the alternative fp32 evaluation orders of the backbone."""
    with torch.backends.mkldnn.flags(enabled=False):
        yield "native-conv", O.backbone(x, P)
    yield "float64-rounded", {k: v.float() for k, v in O.backbone(x.double(), P64).items()}
    yield "channels-last", {k: v.contiguous() for k, v in O.backbone(x.contiguous(memory_format=torch.channels_last), P).items()}


def examine(seed: int, P, P64):
    frame = O.synthetic_frames(1, seed=seed)
    imgs = O.frames_to_chw(frame)
    cfgs = {t: O.OracleCfg(score_thresh=t) for t in THRESHOLDS}
    x, _ = O.preprocess(imgs, cfgs[THRESHOLDS[0]])
    feats = O.backbone(x, P)
    base = {t: O.detect(imgs, P, cfgs[t], features=feats)[0] for t in THRESHOLDS}
    stable, dev = True, {}
    for name, pf in backbone_variants(x, P, P64):
        dev[name] = max(float((pf[k] - feats[k]).abs().max() / feats[k].abs().max()) for k in feats)
        for t in THRESHOLDS:
            r = O.detect(imgs, P, cfgs[t], features=pf)[0]
            if not M.same_discrete_result(base[t], r):
                stable = False
                break
        if not stable:
            break
    margins = {str(t): M.decision_margins(feats, P, cfgs[t]) for t in THRESHOLDS}
    return stable, margins, {str(t): len(base[t]["scores"]) for t in THRESHOLDS}, dev


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=3000)
    ap.add_argument("--candidates", type=int, default=40)
    ap.add_argument("--want", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "e2e_frames.json"))
    args = ap.parse_args()
    P = O.init_params(2020)
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P.items()}
    chosen, log = [], []
    t0 = time.time()
    for s in range(args.first, args.first + args.candidates):
        stable, margins, dets, dev = examine(s, P, P64)
        log.append(dict(seed=s, stable=stable, detections=dets, feature_deviation=dev))
        print(f"seed {s}: stable={stable} detections={dets} [{time.time() - t0:.0f}s]", flush=True)
        if stable:
            chosen.append(dict(seed=s, detections=dets, margins=margins))
        if len(chosen) >= args.want:
            break
    doc = dict(
        note="frames = oracle.planercnn_oracle.synthetic_frames(1, seed); weights = init_params(2020); chosen by oracle/seed_search.py",
        variants=["native-conv", "float64-rounded", "channels-last"], thresholds=list(THRESHOLDS), examined=log, frames=chosen)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(f"wrote {args.out}: {len(chosen)} stable seeds of {len(log)} examined")


if __name__ == "__main__":
    main()
