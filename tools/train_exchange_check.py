#!/usr/bin/env python3
"""Equality check of the training step's gradient exchange forms (tests/test_gpu_training.py runs it as a child process,
because it initialises a process group).

    python tools/train_exchange_check.py --mode world1            # ONE rank on RCCL ("nccl"): segments announced DURING the backward pass
                                                                  # ("force") against the same segments announced behind it ("force-late")
    python tools/train_exchange_check.py --mode world2 [--backend gloo]   # two ranks (gloo: both on this box's one GPU; nccl: two GPUs):
                                                                  # segmented + overlapped ("1") against the monolithic all-reduce ("0")

For each gradient payload (fp32, bf16) two trainers start from the same weights, run `--steps` optimiser steps on the same per-rank batches
and must end with bit-identical parameters and momenta.  The bf16 payload is the sharp case at world 1: its cast overwrites the segment
with what the communication stream saw, so a segment announced before its last weight gradient landed differs.  One JSON line on stdout.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run(rank, world, backend, port, steps, batch, precision, out):
    import torch.distributed as dist

    from train_bench import synthetic_targets
    from bench import build_detector
    from articulation3d_amd.streams import side
    from articulation3d_amd.training import DetectorTrainer
    from articulation3d_amd.utils.synthetic import synthetic_frames
    from articulation3d_amd import parallel

    dev_i = rank % torch.cuda.device_count() if backend == "nccl" else 0
    torch.cuda.set_device(dev_i)
    dev = f"cuda:{dev_i}"
    side(0)  # the package's streams first (streams.py)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    model, _cfg = build_detector(0.5, dev)
    frames = torch.from_numpy(synthetic_frames(batch, seed=2020 + rank)).to(dev)  # (rank-dependent data: the sum is not 2 x one rank's)
    tg = synthetic_targets(batch, 2020 + rank)
    gtb, gtc = [t[0] for t in tg], [t[1] for t in tg]
    forms = ("force", "force-late") if world == 1 else ("1", "0")
    res = {}
    for payload in ("fp32", "bf16"):
        ends = []
        for form in forms:
            tr = DetectorTrainer(model, seed=5, precision=precision, grad_payload=payload, grad_overlap=form)
            parallel.GRAD_STATS.update(steps=0, segments=0, bytes=0, host_s=0.0)
            for _ in range(steps):
                losses, _ = tr.step(frames, gtb, gtc)
            torch.cuda.synchronize()
            ends.append((tr.params.clone(), tr.momentum.clone(), {k: float(v) for k, v in losses.items()}, dict(parallel.GRAD_STATS)))
        (p0, m0, l0, s0), (p1, m1, l1, s1) = ends
        res[payload] = {"params_equal": bool(torch.equal(p0, p1)), "momentum_equal": bool(torch.equal(m0, m1)), "losses_equal": l0 == l1,
                        "finite": bool(torch.isfinite(p0).all()),
                        "segments_per_step": [s0["segments"] // max(steps, 1), s1["segments"] // max(steps, 1)],
                        "payload_bytes_per_step": s0["bytes"] // max(steps, 1), "forms": list(forms)}
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        line = json.dumps({"world": world, "backend": backend, "steps": steps, "batch": batch, "precision": precision, "result": res})
        if out is None:
            print(line)
        else:
            out.put(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="world1", choices=["world1", "world2"])
    ap.add_argument("--backend", default=None)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--precision", default="bf16")
    a = ap.parse_args()
    port = _free_port()
    if a.mode == "world1":
        run(0, 1, a.backend or "nccl", port, a.steps, a.batch, a.precision, None)
        return
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")  # (this parent never touches the GPU)
    q = ctx.Queue()
    procs = [ctx.Process(target=run, args=(r, 2, a.backend or "gloo", port, a.steps, a.batch, a.precision, q)) for r in range(2)]
    for p in procs:
        p.start()
    line = q.get(timeout=1500)
    for p in procs:
        p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    print(line)


if __name__ == "__main__":
    main()
