#!/usr/bin/env python3
"""Dev tool (GPU box): HIP path vs CPU oracle, END TO END, on a range of synthetic-frame seeds -- which seeds give matched
detections on the real hardware, and how large the deviations are.  Used to calibrate oracle/seed_search.py's
perturbation size; the committed test is tests/test_gpu_e2e.py.

    python tools/e2e_explore.py --first 3000 --count 32 --out gpurun_out/e2e_explore.json
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=3000)
    ap.add_argument("--count", type=int, default=32)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "e2e_explore.json"))
    args = ap.parse_args()
    from conftest import make_cfg
    from articulation3d_amd.modeling import build_model
    from oracle import matching as M
    from oracle import planercnn_oracle as O

    P = O.init_params(2020)
    model = build_model(make_cfg(0.0)).eval()
    model.load_state_dict(P, strict=False)
    seeds = list(range(args.first, args.first + args.count))
    frames = np.concatenate([O.synthetic_frames(1, seed=s) for s in seeds])
    res = {}
    for t in (0.5, 0.0):
        model.roi_heads.box_predictor.test_score_thresh = t
        got = []
        for i in range(0, len(seeds), 8):
            out = model.inference_batched(torch.from_numpy(frames[i:i + 8]).cuda(), want_masks=True)
            got += M.gpu_frame_results(out)
        cfg = O.OracleCfg(score_thresh=t)
        rows = []
        for s, f, g in zip(seeds, frames, got):
            o, aux = O.detect(O.frames_to_chw(f[None]), P, cfg, return_aux=True)
            m = M.compare_frame(g, o[0])
            m["seed"] = s
            rows.append(m)
            print(t, json.dumps(m), flush=True)
        res[str(t)] = rows
        print(t, "SUMMARY", json.dumps(M.summarize(rows)), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
