#!/usr/bin/env python3
"""Dev tool (GPU box): box pooler + fc1 of the detector at 64 x 1000 proposals -- fp32 pooled rows + the register-staged wide kernel
against pre-split rows (a3d_roialign_desc.out_h2) + the dual-DMA kernel (a3d_conv_desc.x_h2).  Times per launch, bit equality.
    python tools/fc1_bench.py [--frames 64]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    args = ap.parse_args()
    from bench import build_detector
    from articulation3d_amd import ops
    from articulation3d_amd.utils.synthetic import synthetic_frames

    model, _ = build_detector(0.5, "cuda:0")
    fr = torch.from_numpy(synthetic_frames(args.frames, 2020)).cuda()
    x4 = ops.preprocess_u8hwc(fr, model.pixel_mean, model.pixel_std)
    feats = model.backbone.forward_nhwc(x4)
    pb, _pl, _lv, _pos, pc = model.proposal_generator.forward_batched(feats, (480, 640))
    rh = model.roi_heads
    lv = [feats[f] for f in rh.box_in_features]
    scales = rh.box_pooler.scales
    fc1, fc2 = rh.box_head.fcs[0].packed(), rh.box_head.fcs[1].packed()
    n = int(pc.sum())
    pool32 = lambda: ops.roi_align_fpn(lv, scales, pb, pc, 7, 0, True)
    poolh2 = lambda: ops.roi_align_fpn(lv, scales, pb, pc, 7, 0, True, presplit=True)
    a32, ah2 = pool32(), poolh2()
    lin32 = lambda: ops.linear(ops.keep_amax(a32.view(a32.shape[0], -1), a32), fc1)
    linh2 = lambda: ops.linear(ah2, fc1)
    y32 = lin32()
    v32 = ops.last_conv_variant()
    yh2 = linh2()
    vh2 = ops.last_conv_variant()
    R = pb.shape[1]
    live = torch.cat([torch.arange(int(c)) + b * R for b, c in enumerate(pc.tolist())]).cuda()
    same = bool(torch.equal(y32[live], yh2[live]))
    t = {k: timeit(f) for k, f in (("pool fp32", pool32), ("pool presplit", poolh2), ("fc1 " + v32, lin32), ("fc1 " + vh2, linh2))}
    fl = 2.0 * a32.shape[0] * 12544 * 1024
    for k, v in t.items():
        extra = f"  {fl / v / 1e9:7.1f} fp32-eq TFLOP/s = {3 * fl / v / 1e9 / 2500:.3f} of the 16-bit pipe" if k.startswith("fc1") else f"  {n * 49 * 1024 / v / 1e6:7.1f} GB/s written"
        print(f"{k:34s} {v:7.3f} ms{extra}", flush=True)
    ks = list(t)
    print(f"rows {a32.shape[0]} (live {n}); pooler + fc1: {t[ks[0]] + t[ks[2]]:.3f} -> {t[ks[1]] + t[ks[3]]:.3f} ms; fc1 outputs bit-identical on live rows: {same}", flush=True)


if __name__ == "__main__":
    main()
