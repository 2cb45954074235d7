"""Concurrency between independent branches of a 1-2 frame batch.

A 1-frame pass is not launch-bound but latency-bound ON the GPU: ~200 kernels whose grids fill a fraction of the 256 CUs
while every workgroup still walks its whole k loop.  Independent branches (the FPN output convs, the
RPN levels, the mask / plane / axis heads) therefore run on side HIP streams for 1-2 frame batches.  Measured per batch size
(bench.py --batch b, separate processes, A3D_SMALL_BATCH=0 vs 16): +3 % frames/s at 1 frame, +1 % at 2, -4 % at 4 and 8,
-1..3 % at 16 -- from 4 frames on the branches fill the chip by themselves and the forks only add event waits, hence the
threshold of 2.  (The depth decoder beside the whole ROI branch is a separate, much larger overlap: meta_arch.py, up to 16 frames.)
"""
from __future__ import annotations

import os
from typing import Callable, List, Sequence

import torch

SMALL_BATCH = int(os.environ.get("A3D_SMALL_BATCH", "2"))  # frames (see above); 0 disables
N_SIDE = 3
_side: dict = {}


def queue_setting() -> dict:
    """What decides stream placement in this process: the hardware-queue count of the HIP runtime (GPU_MAX_HW_QUEUES; unset = the
    runtime's default of 4) and whether the package's pool was created before any other stream could take a queue
    (`pool_first`: side() was first called while the process had made no other use of the GPU that this module can see)."""
    v = os.environ.get("GPU_MAX_HW_QUEUES")
    return {"GPU_MAX_HW_QUEUES": v, "pool_created": bool(_side), "pool_first": bool(_side) and not _late_pool,
            "note": "4 queues (default): use the pool first (15.1 | 18.1 ms per 16-image training step otherwise); >= 6 queues: order-free but the "
                    "gradient exchange's cross-stream waits cost 5-6 ms per step (articulation3d_amd/__init__.py)"}


_late_pool = False


def side(i: int, device=None) -> "torch.cuda.Stream":
    """Side stream number i (0 .. N_SIDE - 1, larger i wrap) of the package's ONE pool per device.

    (Round 6 measured the alternative -- more hardware queues, GPU_MAX_HW_QUEUES -- and did not take it: 6+ queues remove the order
    dependence but make the gradient exchange's cross-stream waits cost 5-6 ms per step; articulation3d_amd/__init__.py has the numbers.)
    Why one pool with fixed roles instead of a stream per user.  ROCm maps a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES,
    4 by default) in the order of their FIRST USE, and which queue a stream lands on decides how its kernels interleave with another
    stream's (tools/probes/stream_overlap_matrix.py, train_stream_queues.py: the training step's two side streams as the 5th and 6th
    streams of a process -- behind bench.py's copy stream and three detectors' decoder streams -- ran the 16-image step in 18.0 ms
    instead of 14.9; as the 1st and 2nd they never did).  So the package creates its side streams once, all together and first-used in
    index order right behind the default stream, and every component takes them by role:
        0  the depth decoder beside the ROI branch (meta_arch) | the training step's weight gradients | the gather's count read-back
        1  second branch of a 1-2 frame batch              | the training step's RPN-head stream
        2  third branch of a 1-2 frame batch               | the clip / record copies of a pipelined caller (bench.py) | the training
           step's gradient exchange (parallel.GradientExchange: casts + the collective's enqueue)
    Two roles of one row never run at the same time; if a caller made them, they would serialise on one stream -- slower, never wrong."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev is None:
        dev = torch.cuda.current_device()
    pool = _side.get(dev)
    if pool is None:
        global _late_pool
        import articulation3d_amd as _pkg

        # (the runtime was already up when the package was imported AND the pool comes later still: first-use order is the caller's)
        _late_pool = _late_pool or bool(getattr(_pkg, "_hip_up_at_import", False))
        with torch.cuda.device(dev):
            pool = [torch.cuda.Stream(device=dev) for _ in range(N_SIDE)]
            for st in pool:  # first use, in index order: the hardware queues are taken now, in this order
                with torch.cuda.stream(st):
                    torch.cuda._sleep(1)
            for st in pool:
                st.synchronize()
        _side[dev] = pool
    return pool[i % N_SIDE]


def _stream(i: int) -> "torch.cuda.Stream":
    return side(i)


def _tensors(o):
    if isinstance(o, torch.Tensor):
        yield o
    elif isinstance(o, (list, tuple)):
        for x in o:
            yield from _tensors(x)
    elif isinstance(o, dict):
        for x in o.values():
            yield from _tensors(x)


def run_branches(fns: Sequence[Callable[[], object]], concurrent: bool) -> List[object]:
    """Evaluate independent thunks; with `concurrent` the first runs on the current stream and the others on side streams
    that start behind everything enqueued so far and are joined before returning.  Inputs must stay referenced by the
    caller until the call returns (they are: the join orders any later reuse behind the side streams)."""
    if not concurrent or len(fns) < 2 or not torch.cuda.is_available():
        return [f() for f in fns]
    main = torch.cuda.current_stream()
    ready = torch.cuda.Event()
    ready.record(main)
    outs: List[object] = [None] * len(fns)
    done = []
    for i in range(1, len(fns)):
        s = _stream(i)
        with torch.cuda.stream(s):
            s.wait_event(ready)
            outs[i] = fns[i]()
            e = torch.cuda.Event()
            e.record(s)
            done.append(e)
    outs[0] = fns[0]()
    for e in done:
        main.wait_event(e)
    for o in outs[1:]:
        for t in _tensors(o):
            t.record_stream(main)  # allocated on a side stream, consumed (and later freed) on the main one
    return outs
