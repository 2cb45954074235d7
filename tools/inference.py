#!/usr/bin/env python3
"""The reference's tools/inference.py on this package: per-clip articulation prediction.

    python tools/inference.py --config configs/planercnn_inference.yaml --input <frames> --output out/ [--conf-threshold 0.7]

Same stages as the reference script (tools/inference.py:171-250): build the model from the YAML, detect every frame,
build the `create_instances` records, `track_planes`, `optimize_planes(preds, planes, '3dc')` -- but the frame loop is the
batched, frame-sharded `pipeline.detect_clip` (one RCCL all-gather per clip when launched with torchrun) and the
optimiser's sweeps run on the GPU.  `random.seed(2020)` / `np.random.seed(2020)` as in the reference (:172-173).

Input (this image has no video decoder: cv2 / imageio are absent, so .mp4 is refused with a clear message):
  * a .npy / .npz file with uint8 frames [F,H,W,3] in RGB order (what imageio would hand over), or
  * a directory of .png / .jpg frames (decoded with PIL, sorted by name), or
  * `synthetic:N` / `synthetic:NxHxW` -- N seeded random frames (plumbing check), optionally at a source size H x W.
The frames go to the GPU as they come from the reader (uint8 RGB, any size): the reference's `cv2.resize(im, (640, 480))` and
BGR flip (:216-218) run on the device, fused with the normalisation in front of the stem (a3d_preprocess_resize_u8).
Output: <output>/predictions.json -- per frame the optimised detections (bbox xyxy, score, class, plane, rotation /
translation axis, RLE mask) and <output>/tracks.json (tracked planes, has_rot, consensus axis).  The 2-D / 3-D
visualisations of the reference (imageio video, pytorch3d meshes) are out of scope.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def count_frames(path: str) -> int:
    """Number of frames of a clip without decoding it (npy header / directory listing)."""
    if path.startswith("synthetic:"):
        return int(path.split(":")[1].split("x")[0])
    if path.endswith(".npy"):
        return int(np.load(path, mmap_mode="r").shape[0])
    if path.endswith(".npz"):
        z = np.load(path)
        return int(z[list(z.keys())[0]].shape[0])
    if os.path.isdir(path):
        return len([n for n in os.listdir(path) if n.lower().endswith((".png", ".jpg", ".jpeg"))])
    return 1


def read_frames(path: str, lo: int = 0, hi: int = None) -> np.ndarray:
    """-> uint8 [hi-lo,Hs,Ws,3] RGB at the SOURCE size (no host-side resize): frames lo .. hi-1 of the clip (default: all).  Under
    torchrun every rank reads only its own block (memory-mapped .npy, per-file image decode)."""
    from PIL import Image

    sl = slice(lo, hi)

    if path.startswith("synthetic:"):
        from articulation3d_amd.utils.synthetic import synthetic_frames

        spec = path.split(":")[1].split("x")
        n, h, w = int(spec[0]), (int(spec[1]) if len(spec) == 3 else 480), (int(spec[2]) if len(spec) == 3 else 640)
        return synthetic_frames(n, h=h, w=w)[sl][..., ::-1].copy()  # generated BGR -> RGB, flipped back on the device
    if path.endswith((".mp4", ".avi", ".mov")):
        raise SystemExit("no video decoder in this environment (cv2 / imageio absent): pass a .npy of RGB frames or a directory of images")
    if path.endswith(".npy"):
        frames = np.load(path, mmap_mode="r")[sl]
    elif path.endswith(".npz"):
        z = np.load(path)
        frames = z[list(z.keys())[0]][sl]
    elif os.path.isdir(path):
        every = sorted(n for n in os.listdir(path) if n.lower().endswith((".png", ".jpg", ".jpeg")))
        names = every[sl]
        if names:
            frames = np.stack([np.asarray(Image.open(os.path.join(path, n)).convert("RGB")) for n in names])
        else:  # an empty block (trailing ranks when F <= ceil(F / world) * (world - 1)): zero frames at the clip's size
            if not every:
                raise SystemExit(f"{path}: no .png / .jpg frames")
            w, h = Image.open(os.path.join(path, every[0])).size
            frames = np.zeros((0, h, w, 3), np.uint8)
    else:
        frames = np.asarray(Image.open(path).convert("RGB"))[None]
    assert frames.ndim == 4 and frames.shape[3] == 3 and frames.dtype == np.uint8, "frames must be uint8 [F,H,W,3]"
    return np.ascontiguousarray(frames)


def main():
    random.seed(2020)
    np.random.seed(2020)
    ap = argparse.ArgumentParser(description="Articulation prediction on a clip (MI355X).")
    ap.add_argument("--config", required=True)
    ap.add_argument("--input", required=True)
    ap.add_argument("--output", required=True)
    ap.add_argument("--conf-threshold", default=0.7, type=float)
    ap.add_argument("--batch", default=32, type=int, help="frames per detector batch per GPU")
    ap.add_argument("--calibrate-bn", action="store_true", help="random-init weights only: estimate batch-norm statistics on the clip")
    ap.add_argument("--random-init", action="store_true",
                    help="do NOT load cfg.MODEL.WEIGHTS: explicit random initialisation (plumbing runs; a missing checkpoint is otherwise an error)")
    ap.add_argument("--dist-backend", default="nccl", help="torchrun only: nccl (= RCCL over xGMI) or gloo")
    ap.add_argument("--no-precision-audit", action="store_true",
                    help="skip the load-time audit of the default arithmetic (PlaneRCNN.audit_precision: every fp16x2 layer checked against its "
                         "bf16x3 evaluation on the clip's first frames; layers that leave the format's window are pinned to bf16x3)")
    ap.add_argument("opts", nargs=argparse.REMAINDER, default=[], help="KEY VALUE config overrides, e.g. MODEL.DEVICE cuda:0")
    args = ap.parse_args()

    # one process per GPU under torchrun (frames sharded by pipeline.detect_clip, ONE all-gather of the records per clip):
    # rank / device / process group are set up before anything touches the GPU; rank 0 alone tracks, optimises and writes
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0:
        os.makedirs(args.output, exist_ok=True)

    from articulation3d_amd.config import get_cfg, get_planercnn_cfg_defaults
    from articulation3d_amd.pipeline import detect_clip
    from articulation3d_amd.utils import rle
    from articulation3d_amd.utils.arti_vis import PlaneRCNN_Branch
    from articulation3d_amd.utils.opt_utils import optimize_planes, track_planes

    cfg = get_cfg()
    get_planercnn_cfg_defaults(cfg)
    cfg.merge_from_file(args.config)
    if args.opts:
        cfg.merge_from_list(args.opts)
    if not str(cfg.MODEL.DEVICE).startswith("cuda"):
        raise SystemExit(f"MODEL.DEVICE={cfg.MODEL.DEVICE!r}: this package is the MI355X (HIP) implementation of the detection path and has "
                         "no CPU fallback; the CPU statement of the same graph lives in oracle/ as the test checker")
    if world > 1:
        import torch.distributed as dist

        n_dev = torch.cuda.device_count()  # (does not initialise HIP)
        cfg.MODEL.DEVICE = f"cuda:{local_rank % max(n_dev, 1)}"
        torch.cuda.set_device(local_rank % max(n_dev, 1))
        from articulation3d_amd.streams import side

        side(0)  # the package's side streams take their hardware queues now, in front of the collective library's (articulation3d_amd/streams.py)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(cfg.MODEL.DEVICE))
        else:
            dist.init_process_group(backend=args.dist_backend)
    if args.random_init:
        torch.manual_seed(2020)  # every rank (and every run) draws the same initialisation
    branch = PlaneRCNN_Branch(cfg, load_weights=not args.random_init)
    model = branch.predictor.model
    from articulation3d_amd.parallel import shard_range

    n_frames = count_frames(args.input)
    lo, hi = shard_range(n_frames, rank, world)
    frames_rgb = read_frames(args.input, lo, hi)  # this rank's block only (rank 0 reads the rest later, for the optimiser)
    if args.calibrate_bn:
        # the statistics come from the FIRST two frames of the clip on EVERY rank (rank 0's own block may hold only one of them)
        n_head = min(2, n_frames)
        frames_rgb_head = frames_rgb[:n_head] if lo == 0 and hi - lo >= n_head else read_frames(args.input, 0, n_head)
        from articulation3d_amd import ops
        from articulation3d_amd.utils.synthetic import calibrate_batchnorm

        _x4, head = ops.preprocess_resize_u8(torch.from_numpy(np.ascontiguousarray(frames_rgb_head)).to(model.device), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), want_u8=True)
        calibrate_batchnorm(model, head.flip(-1).contiguous())  # resized RGB -> BGR uint8
    if not args.no_precision_audit:
        # once per loaded checkpoint, on the FIRST two frames of the clip on every rank (every rank must pin the same layers: a frame's
        # result may not depend on the rank that detects it)
        n_head = min(2, n_frames)
        head_rgb = frames_rgb[:n_head] if lo == 0 and hi - lo >= n_head else read_frames(args.input, 0, n_head)
        audit = model.audit_precision(torch.from_numpy(np.ascontiguousarray(head_rgb)).to(model.device), source_rgb=True)
        if rank == 0:
            worst = max((r["max_ratio"] for r in audit.rows), default=0.0)
            print(f"precision audit: {len(audit.rows)} fp16x2 layer launches checked against bf16x3 on {n_head} frame(s), worst err / bound {worst:.3f}; "
                  f"pinned to bf16x3: {model.pinned_layers() or 'none'}")
    preds = detect_clip(model, frames_rgb, batch=args.batch, conf_threshold=args.conf_threshold, source_rgb=True, num_frames=n_frames)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:  # every rank holds the gathered detections; the temporal stage and the files are rank 0's
            return
        frames_rgb = read_frames(args.input)  # the optimiser's sweeps look at the whole clip
    planes = track_planes(preds)
    opt_preds = optimize_planes(preds, planes, "3dc", frames=frames_rgb)

    out = []
    for i, p in enumerate(opt_preds):
        n = len(p.pred_boxes)
        out.append({"frame": i, "instances": [{
            "bbox": [float(v) for v in p.pred_boxes.tensor[k]], "score": float(p.scores[k]), "category_id": int(p.pred_classes[k]),
            "pred_plane": [float(v) for v in p.pred_planes[k]], "pred_rot_axis": [float(v) for v in p.pred_rot_axis[k]],
            "pred_tran_axis": [float(v) for v in p.pred_tran_axis[k]],
            "segmentation": rle.encode(np.asfortranarray(p.pred_masks[k].numpy().astype(np.uint8)))} for k in range(n)]})
    with open(os.path.join(args.output, "predictions.json"), "w") as f:
        json.dump(out, f)
    tracks = {cat: [{"frames": sorted(t["ids"]), "has_motion": bool(t.get("has_rot", False)),
                     "consensus_axis": (np.asarray(t["std_axis"]).tolist() if "std_axis" in t else None)} for t in ts]
              for cat, ts in planes.items()}
    with open(os.path.join(args.output, "tracks.json"), "w") as f:
        json.dump(tracks, f)
    kept = sum(len(p.pred_boxes) for p in opt_preds)
    print(f"{len(frames_rgb)} frames, {kept} detections above {args.conf_threshold}, "
          f"{len(planes['rot'])} rotation / {len(planes['trans'])} translation tracks -> {args.output}")


if __name__ == "__main__":
    main()
