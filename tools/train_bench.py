#!/usr/bin/env python3
"""Throughput of the training step (BASELINE configs[4]: train_net.py step1_bbox.yaml) on MI355X.

    python tools/train_bench.py [--gpus N --steps K --warmup W --batch 2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py --gpus N ...

One step = forward + losses + backward + gradient all-reduce (N > 1, RCCL) + SGD on `--batch` synthetic 480x640 frames per
GPU (the reference trains 16 images over 8 GPUs = 2 per GPU, step1_bbox.yaml:40).  Same timing discipline as bench.py:
W untimed steps, K timed steps between barrier + synchronize, max over ranks, ONE JSON line from rank 0.  This is a
secondary metric (the headline metric of BASELINE.json is bench.py's frames/s through the detector).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synthetic_targets(n, seed, h=480, w=640):
    """2-6 random ground-truth boxes per frame, classes in {0,1} (same generator as the CPU oracle's)."""
    rng = np.random.default_rng(seed + 7)
    out = []
    for _ in range(n):
        g = int(rng.integers(2, 7))
        bw, bh = rng.uniform(40, 320, g), rng.uniform(40, 260, g)
        x1, y1 = rng.uniform(0, w - bw), rng.uniform(0, h - bh)
        boxes = np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
        out.append((torch.from_numpy(boxes), torch.from_numpy(rng.integers(0, 2, g).astype(np.int64))))
    return out


def train_leg(dev, batch, steps, warmup, precision="bf16", storage=None, cpu_baseline=False, layers=False, rank=0, world=1, dist=None, model=None):
    """W untimed + K timed training steps on `batch` synthetic frames per GPU, then ONE fully instrumented step for the roofline object.
    Returns the result dict of the JSON line (bench.py embeds it as `train_step`; this file's main() prints it)."""
    from bench import build_detector
    from articulation3d_amd.training import DetectorTrainer
    from articulation3d_amd.utils.synthetic import synthetic_frames

    if model is None:
        model, _cfg = build_detector(0.5, dev)
    tr = DetectorTrainer(model, seed=2020 + rank, precision=precision, storage=storage)
    B = batch
    frames = torch.from_numpy(synthetic_frames(B, seed=2020 + rank)).to(dev)
    tg = synthetic_targets(B, 2020 + rank)
    gtb, gtc = [t[0] for t in tg], [t[1] for t in tg]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from articulation3d_amd import parallel as _par

    for _ in range(warmup):
        tr.step(frames, gtb, gtc)
    barrier()
    # Two consecutive timed blocks of `steps` steps, the faster one reported (`timing` in the line): a 5 ms step of ~350 launches is at the
    # mercy of one host hiccup per block (allocator growth behind bench.py's empty_cache, a collector pause: 7.7 against 5.2-5.5 ms seen once
    # in four driver-style runs); every rank takes the same block (the MAX over ranks is formed per block).
    blocks = []
    for _rep in range(2):
        _par.GRAD_STATS.update(steps=0, segments=0, bytes=0, host_s=0.0)
        t0 = time.perf_counter()
        for _ in range(steps):
            losses, _ = tr.step(frames, gtb, gtc)
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device="cpu" if dist.get_backend() == "gloo" else dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        blocks.append((el, dict(_par.GRAD_STATS)))
    elapsed, gstats = min(blocks, key=lambda b: b[0])
    # ---- roofline of the step: ONE fully instrumented step after the timed loop -- HIP events, on the launch stream,
    # around every conv / linear launch (forward and data gradients: ops.conv2d) and every weight-gradient launch (train_ops.conv_wgrad).
    # Executed FLOPs: Winograd layers 16 / 36 of the direct count; the fp32-grade step issues SIX bf16 MFMA products per multiply-add.
    from articulation3d_amd import ops
    from bench import PIPE_FLOPS_PER_FMA, PIPE_PEAK, dominant_roofline, kernel_sums

    # The instrumented step runs everything on the main stream (the timed steps above keep the weight gradients and the RPN head's backward
    # on their side streams): an event bracket then times one launch alone, not the span it shares the GPU with another stream.
    side, tr._wg_stream = tr._wg_stream, None
    rside, tr._rpn_stream = tr._rpn_stream, None
    ops.CONV_TIMING = []
    tr.step(frames, gtb, gtc)
    barrier()
    events, ops.CONV_TIMING = ops.CONV_TIMING, None
    tr._wg_stream, tr._rpn_stream = side, rside
    if layers and rank == 0:
        for name, fl, a, b, shape, ex, pipe, _st in events:
            ms = a.elapsed_time(b)
            print(f"{name[:40]:40s} {shape:52s} {ex / 1e9:9.2f} GF {ms:8.3f} ms {ex / ms / 1e9 if ms > 0 else 0:8.1f} TF/s", file=sys.stderr)
    per = kernel_sums(events)
    step_sec = elapsed / steps
    roofline = dominant_roofline(per, step_sec)
    issued = sum(v[3] * PIPE_FLOPS_PER_FMA[v[4]] for v in per.values())
    pipes = {v[4] for v in per.values() if v[3]}
    step_peak = max(PIPE_PEAK[q] for q in pipes)
    roofline["source"] = ("one fully instrumented step after the timed loop, run on ONE stream (every conv / linear / weight-gradient launch bracketed by HIP "
                          "events, each kernel alone on the chip); `value` / `whole_step` use the timed steps, whose side streams overlap")
    roofline["gemm_kernels_ms_per_step"] = round(1e3 * sum(v[1] for v in per.values()), 3)
    roofline["whole_step"] = {"executed_tflop_per_step": round(issued / 1e12, 3), "achieved": round(issued / step_sec / 1e12, 2), "peak": step_peak,
                              "unit": "TFLOP/s", "frac": round(issued / step_sec / 1e12 / step_peak, 4),
                              "fp32_equivalent_tflop_per_step": round(sum(v[3] for v in per.values()) / 1e12, 3),
                              "note": "all matrix-pipe FLOPs the step executes (forward + data gradients + weight gradients of the trainable layers, frozen "
                                      "stem / res2 forward included) over the WALL time of a step, against the dense peak of the pipe the step runs on"}
    roofline["all_gemm_kernels"] = {k: {"pipe": v[4], "ms_per_step": round(1e3 * v[1], 3), "launches_per_step": v[2],
                                        "frac_of_pipe_peak": round(v[3] * PIPE_FLOPS_PER_FMA[v[4]] / v[1] / 1e12 / PIPE_PEAK[v[4]], 4) if v[3] else None}
                                    for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])}
    result = {
        "metric": "images/sec through the step1_bbox training step at 480x640", "value": round(B * world * steps / elapsed, 2),
        "unit": "images/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * elapsed / steps, 3),
        "timing": f"the faster of two consecutive blocks of {steps} steps (ms per step of both: {', '.join('%.3f' % (1e3 * b[0] / steps) for b in blocks)})",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "bf16": "bf16 (fp32 accumulate, fp32 master weights)",
                                                                         "bf16x3": "f32 via exact 3-way bf16 operand split (forward / data gradients of the non-Winograd layers)"}[precision], "data": "synthetic",
        "config": {"workload": "BASELINE configs[4]: Faster R-CNN training step of step1_bbox.yaml (ResNet50-FPN, FREEZE_AT 2, RPN + box head "
                               "losses, SGD momentum), random-init weights with calibrated BN, synthetic frames and boxes (arithmetic: see `dtype`)",
                   "precision": precision, "storage": tr.storage,
                   "images_per_gpu": B, "global_batch": B * world, "trainable_parameters": int(tr.params.numel()),
                   "gradient_exchange": ("four segments of the flat gradient buffer, each all-reduced over RCCL under the rest of the backward pass "
                                         "(parallel.GradientExchange)") if world > 1 else "none (1 GPU)"},
        "losses_last_step": {k: round(float(v), 5) for k, v in losses.items()},
        "roofline": roofline,
    }
    if world > 1:  # (the exchange as it ran in the timed steps; what it leaves exposed is measured at N = 1 by exchange_probe)
        result["collective"] = {"segments": gstats["segments"] // steps, "bytes": gstats["bytes"] // steps,
                                "host_ms_per_step": round(1e3 * gstats["host_s"] / steps, 4), "payload": tr.grad_payload, "overlap": tr.grad_overlap}
    if cpu_baseline and rank == 0:
        from oracle import planercnn_oracle as O, train_oracle as TO

        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = max(1, min(avail, 64))
        torch.set_num_threads(cores)
        P = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        imgs = O.frames_to_chw(synthetic_frames(B, seed=2020))
        t0 = time.perf_counter()
        TO.loss_and_grads(imgs, tg, P, O.OracleCfg(), TO.TrainCfg(), gen=torch.Generator().manual_seed(1))
        dt = time.perf_counter() - t0
        result["cpu_baseline"] = {"value": round(B / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
                                  "sample": f"one forward+backward of the autograd oracle on {B} images (no optimiser step)",
                                  "caveat": "a stated baseline, never the target (oracle/train_oracle.py: torch.autograd over the CPU oracle)"}
        result["gpu_over_cpu"] = round(result["value"] / result["cpu_baseline"]["value"], 1)
    return result


def exchange_probe(dev, model, batch, steps, warmup, precision="bf16", dist_module=None):
    """What the N > 1 step's gradient exchange leaves EXPOSED, measured on one GPU: the step with every segment collective of
    parallel.GradientExchange issued on a ONE-rank RCCL group (`grad_overlap="force"`: segment flushes, events, casts, all_reduce on RCCL's
    stream, the optimiser's wait) against the same step without any exchange.  A one-rank all-reduce moves no bytes over xGMI -- the figure
    is the launch / synchronisation cost of the overlapped form, not the transfer time (DESIGN.md section 6 prices that from link rates).
    dist_module: torch.distributed when the caller already holds a process group; otherwise a one-rank nccl group is created and destroyed."""
    import socket

    import torch.distributed as dist

    from articulation3d_amd import parallel
    from articulation3d_amd.training import DetectorTrainer
    from articulation3d_amd.utils.synthetic import synthetic_frames

    own = not dist.is_initialized()
    # RCCL prints a version banner to STDOUT (file descriptor 1) when its communicator is created, i.e. at the first collective: the bench
    # contract is ONE JSON line there, so descriptor 1 points at stderr for the duration of the probe.
    sys.stdout.flush()
    keep_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        return _exchange_probe(dev, model, batch, steps, warmup, precision, dist, own)
    finally:
        sys.stdout.flush()
        import ctypes

        ctypes.CDLL(None).fflush(None)  # (the banner sits in the C library's stdout buffer: push it out while descriptor 1 is still stderr)
        os.dup2(keep_fd, 1)
        os.close(keep_fd)


def _exchange_probe(dev, model, batch, steps, warmup, precision, dist, own):
    import socket

    from articulation3d_amd import parallel
    from articulation3d_amd.training import DetectorTrainer
    from articulation3d_amd.utils.synthetic import synthetic_frames

    if own:
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device(dev))
    if dist.get_world_size() != 1 or dist.get_backend() != "nccl":
        return None
    frames = torch.from_numpy(synthetic_frames(batch, seed=2020)).to(dev)
    tg = synthetic_targets(batch, 2020)
    gtb, gtc = [t[0] for t in tg], [t[1] for t in tg]
    ms = {}
    for form in ("0", "force", "0", "force"):  # (interleaved repeats: the better of two per form)
        tr = DetectorTrainer(model, seed=2020, precision=precision, grad_overlap=form)
        for _ in range(warmup):
            tr.step(frames, gtb, gtc)
        torch.cuda.synchronize()
        parallel.GRAD_STATS.update(steps=0, segments=0, bytes=0, host_s=0.0)
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(frames, gtb, gtc)
        torch.cuda.synchronize()
        ms[form] = min(ms.get(form, 1e9), 1e3 * (time.perf_counter() - t0) / steps)
        stats = dict(parallel.GRAD_STATS)
        del tr
    if own:
        dist.destroy_process_group()
    return {"segments": stats["segments"] // steps, "bytes": stats["bytes"] // steps, "exposed_ms": round(ms["force"] - ms["0"], 4),
            "ms_per_step_with_exchange": round(ms["force"], 4), "ms_per_step_without": round(ms["0"], 4),
            "host_ms_per_step": round(1e3 * stats["host_s"] / steps, 4), "payload": "bf16" if precision == "bf16" else "fp32",
            "form": "four segments in backward-completion order (box head | res5 + FPN + RPN head | res4 | res3), each cast + all_reduce(async) on the "
                    "communication stream the moment its last weight gradient is enqueued; ONE-rank RCCL group on this box",
            "note": "exposed_ms = step with the exchange - step without, same process, best of two interleaved runs of 10 steps each"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2, help="images per GPU per step (reference: IMS_PER_BATCH 16 over 8 GPUs)")
    ap.add_argument("--dist-backend", default="nccl")
    ap.add_argument("--storage", default=None, choices=["fp32", "bf16"], help="res3-res5 activations / gradients in HBM (default: bf16 in the bf16 step)")
    ap.add_argument("--precision", default="bf16x3", choices=["fp32", "bf16", "bf16x3"],
                    help="bf16 = autocast arithmetic (bf16 MFMA, fp32 accumulate); bf16x3 = fp32-grade 3-way bf16 split on the non-Winograd layers")
    ap.add_argument("--layers", action="store_true", help="per-launch table of the instrumented step on stderr (developer tool)")
    ap.add_argument("--cpu-baseline", action="store_true", help="also time ONE oracle step (autograd on the host cores)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("train_bench.py needs an MI355X (HIP) device")
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    from articulation3d_amd.streams import side

    side(0)  # the package's side streams take their hardware queues now, in front of the collective library's (articulation3d_amd/streams.py)
    dev = f"cuda:{local_dev}"
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend=args.dist_backend)
    result = train_leg(dev, args.batch, args.steps, args.warmup, precision=args.precision, storage=args.storage,
                       cpu_baseline=args.cpu_baseline and world == 1, layers=args.layers, rank=rank, world=world, dist=dist)
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
