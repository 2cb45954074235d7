"""Frame-sharded detection across the GPUs of one node.

Frames are independent until the temporal tracker (pkg/utils/opt_utils.py:1156-1208), so rank r detects the
contiguous block [r*F/G, (r+1)*F/G) of the clip with replicated weights and no data-path collective; the only
exchange is ONE all-gather per batch of the fixed-size detection records (include/a3d.h a3d_pack_desc) --
the payload `create_instances` would build (pkg/utils/arti_vis.py:152-194) -- so that every rank (rank 0 in
practice) holds the whole clip's detections in temporal order for the host-side optimiser.  The reference's
analogue is `comm.gather` of pickled predictions (pkg/evaluation/arti_evaluation.py:195-199).
Backend: torch.distributed "nccl" (= RCCL over xGMI on ROCm), "gloo" on CPU for tests.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(num_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of rank `rank`; concatenation in rank order restores temporal order."""
    per = (num_frames + world - 1) // world
    lo = min(rank * per, num_frames)
    return lo, min(lo + per, num_frames)


def gather_records(records: torch.Tensor, rec_count: torch.Tensor, rows: Optional[int] = None):
    """records [B,R,F] fp32, rec_count [B] int32 (this rank's block) -> ([G*rows,R,F], [G*rows]) on every rank
    (rows = B when not given: see gather_records_async for the equal-block contract)."""
    return gather_records_async(records, rec_count, rows).wait()


class GatherHandle:
    """Result of gather_records_async.  Two ways to collect it, both ordered on the CURRENT stream (no host wait beyond the counts):
      wait_compact() -> (rows [N, F], counts [G*rows] int32): the live records of the whole job, frame-major in rank order -- what
                        travelled; frame s of rank r is slot r*rows + s, its records start at counts[:slot].sum();
      wait()         -> (all_records [G*rows, R, F], all_counts [G*rows]): the dense slot layout (dead slots zero)."""

    def __init__(self, finish, dense_shape):
        self._finish, self._dense_shape, self._res = finish, dense_shape, None

    def wait_compact(self):
        if self._res is None:
            self._res, self._finish = self._finish(), None
        return self._res

    def wait(self):
        rows, counts = self.wait_compact()
        n, R, F = self._dense_shape
        return scatter_compact(rows, counts, R), counts


def _index_to(idx, device) -> torch.Tensor:
    """A host index list on `device` WITHOUT a host-blocking copy: a pageable source would make the copy wait for everything the
    stream holds (the next batch's kernels) before the host continues; pinned + non_blocking only orders it on the stream."""
    t = torch.tensor(idx, dtype=torch.int64)
    if torch.device(device).type != "cuda":
        return t
    return t.pin_memory().to(device, non_blocking=True)


def compact_index(counts, R: int, device) -> torch.Tensor:
    """Flat slot-row indices (slot * R + r, r < count[slot]) of the live records, frame-major.  counts: host list of ints."""
    return _index_to([s * R + r for s, c in enumerate(counts) for r in range(min(int(c), R))], device)


def scatter_compact(rows: torch.Tensor, counts: torch.Tensor, R: int) -> torch.Tensor:
    """Live rows [N, F] + per-slot counts -> dense [slots, R, F] (zeros in dead slots)."""
    n = counts.numel()
    dense = rows.new_zeros((n * R, rows.shape[1]))
    if rows.shape[0]:
        dense[compact_index(counts.tolist(), R, rows.device)] = rows
    return dense.view(n, R, rows.shape[1])


class _LocalHandle(GatherHandle):
    """Single process: nothing travels.  wait() hands the block back as it is; wait_compact() compacts it (one host read of the counts)."""

    def __init__(self, records, rec_count):
        self._rec, self._cnt = records, rec_count

    def wait(self):
        return self._rec, self._cnt

    def wait_compact(self):
        B, R, F = self._rec.shape
        idx = compact_index(self._cnt.tolist(), R, self._rec.device)
        return self._rec.reshape(B * R, F)[idx], self._cnt


_verified_blocks = set()
# measurement record of the exchanges (bench.py resets and reads it): collectives finished, bytes one rank RECEIVED over both phases,
# host seconds spent inside GatherHandle.wait_compact
STATS = {"calls": 0, "bytes": 0, "host_s": 0.0}


def gather_records_async(records: torch.Tensor, rec_count: torch.Tensor, rows: Optional[int] = None) -> GatherHandle:
    """All-gather of one batch's detection records in two phases -- counts first, then only the LIVE records.

    The slot layout is [frames, R = 100 slots, F = 798 floats] per rank, but a frame holds D ~ 4 detections: shipping the slots is
    20 MB per rank and step to every rank (2.6 GB of dense records for a 1024-frame clip on the receiving side), the live records
    under 1 MB (counts, then payload).
      phase 1 (here, asynchronous on RCCL's stream, under the next batch's kernels): `rows + 1` int32 per rank = the block's frame
              count and its per-frame detection counts, zero-padded to `rows` frames;
      phase 2 (at wait time): every rank compacts its live records, pads them to the largest per-rank total -- known to all from
              phase 1 -- and one `all_gather_into_tensor` moves them; the receiver drops the padding.
    `rows` = frames per rank block, the same on every rank (ceil(F / world) for a sharded clip: `shard_range` hands out uneven
    blocks whenever F % world != 0 -- 10 frames on 4 ranks: 3,3,3,1 -- and shorter blocks are padded with empty frames).  A block
    LARGER than `rows` cannot travel: every rank learns that from phase 1 and all of them raise together (a local raise before the
    collective would leave the others hanging in RCCL).  Without `rows` the caller promises equal blocks; the first call per block
    shape verifies that with one small fixed-size collective and raises on every rank when they differ."""
    if not (dist.is_available() and dist.is_initialized()):
        return _LocalHandle(records, rec_count)
    G = dist.get_world_size()
    rec, cnt = records.contiguous(), rec_count.contiguous().to(torch.int32)
    B, R, F = rec.shape
    gloo = dist.get_backend() == "gloo"
    back = rec.device if (rec.is_cuda and gloo) else None  # test rigs (several ranks on one GPU): gloo moves host memory
    if back is not None:
        rec, cnt = rec.cpu(), cnt.cpu()
    dev = rec.device
    if rows is None:
        key = (B, R, F)
        if key not in _verified_blocks:  # fixed-size exchange: well-formed whatever the blocks are
            sizes = torch.empty((G,), device=dev, dtype=torch.int64)
            dist.all_gather_into_tensor(sizes, torch.tensor([B], device=dev, dtype=torch.int64))
            sizes = sizes.tolist()
            if len(set(sizes)) != 1:
                raise ValueError(f"gather_records: ranks hold blocks of {sizes} frames; pass rows=ceil(F/world) to pad them")
            _verified_blocks.add(key)
        rows = B
    # ---- phase 1: [frames in the block, count[0..rows)] per rank
    hdr = torch.zeros((rows + 1,), device=dev, dtype=torch.int32)
    hdr[0] = B
    nb = min(B, rows)
    hdr[1:1 + nb] = cnt[:nb]
    all_hdr = torch.empty((G * (rows + 1),), device=dev, dtype=torch.int32)
    hdr_host, ev = None, None
    if rec.is_cuda:
        w1 = dist.all_gather_into_tensor(all_hdr, hdr, async_op=True)
        # the counts reach the host through a side stream that waits for the collective only -- not for whatever the caller
        # has enqueued on its stream in the meantime (the next batch)
        side = _side_stream(dev)
        hdr_host = torch.empty((G * (rows + 1),), dtype=torch.int32, pin_memory=True)
        with torch.cuda.stream(side):
            w1.wait()
            hdr_host.copy_(all_hdr, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
    else:
        dist.all_gather_into_tensor(all_hdr, hdr)

    keep_alive = [hdr, all_hdr, rec, cnt]  # referenced until the exchange has been collected

    def finish():
        import time as _time

        _t0 = _time.perf_counter()
        try:
            return _finish()
        finally:
            STATS["calls"] += 1
            STATS["host_s"] += _time.perf_counter() - _t0

    def _finish():
        keep_alive.clear()
        if ev is not None:
            ev.synchronize()
            h = hdr_host.view(G, rows + 1)
        else:
            h = all_hdr.view(G, rows + 1)
        sizes = h[:, 0].tolist()
        if max(sizes) > rows:  # seen by every rank: all of them raise
            raise ValueError(f"gather_records: ranks hold blocks of {sizes} frames but rows={rows}; pass rows=ceil(F/world)")
        counts = h[:, 1:].contiguous()                      # [G, rows]
        live = [int(v) for v in counts.clamp(max=R).sum(1).tolist()]
        mx = max(live)
        all_cnt = counts.reshape(-1).to(dev, non_blocking=True)  # (pinned source on the device path: no host wait)
        STATS["bytes"] += G * (rows + 1) * 4 + G * mx * F * rec.element_size()
        if mx == 0:
            out = (rec.new_zeros((0, F)), all_cnt)
        else:
            mine = compact_index(counts[dist.get_rank()].tolist()[:B], R, dev)
            send = rec.new_zeros((mx, F))
            if mine.numel():
                send[: mine.numel()] = rec.view(B * R, F)[mine]
            got = rec.new_empty((G * mx, F))
            if rec.is_cuda:
                dist.all_gather_into_tensor(got, send, async_op=True).wait()  # (the current stream waits; the host does not)
            else:
                dist.all_gather_into_tensor(got, send)
            keep = _index_to([g * mx + i for g in range(G) for i in range(live[g])], dev)
            out = (got[keep], all_cnt)
        if back is not None:
            out = (out[0].to(back), out[1].to(back))
        return out

    return GatherHandle(finish, (G * rows, R, F))


def _side_stream(device):
    from .streams import side  # (the package's one pool of side streams: streams.py)

    return side(0, device)


def allreduce_gradients(flat_grads: torch.Tensor, group=None, payload: str = "fp32") -> float:
    """Data-parallel training (SURVEY.md 8e "training", 8f-1): ONE sum all-reduce of the trainer's flat gradient buffer
    (replaces DistributedDataParallel's bucketed all-reduce behind tools/train_net.py:110).  Returns the factor the
    optimiser applies to turn the sum into DDP's mean (1 / world size); 1.0 when not distributed.

    payload="bf16" (BASELINE configs[4]: bf16 gradient all-reduce over xGMI; torch DDP's `bf16_compress_hook`): the gradient is
    divided by the world size and rounded to bf16 BEFORE the collective, summed in bf16 and widened back into `flat_grads` -- half
    the bytes on the links (82 MB instead of 164 MB for the 41 M trainable parameters); the returned factor is then 1.0."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world == 1:
        return 1.0
    if payload == "bf16":
        n = flat_grads.numel()
        if flat_grads.is_cuda:
            from . import _lib

            g16 = torch.empty(n, device=flat_grads.device, dtype=torch.bfloat16)
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(_lib.lib().a3d_f32_to_bf16_scaled(flat_grads.data_ptr(), g16.data_ptr(), n, 1.0 / world, st), "a3d_f32_to_bf16_scaled")
            dist.all_reduce(g16, group=group)
            _lib.check(_lib.lib().a3d_bf16_to_f32(g16.data_ptr(), flat_grads.data_ptr(), n, st), "a3d_bf16_to_f32")
        else:  # host tensors (gloo test rigs): the same two casts
            g16 = (flat_grads * (1.0 / world)).to(torch.bfloat16)
            dist.all_reduce(g16, group=group)
            flat_grads.copy_(g16.to(torch.float32))
        return 1.0
    assert payload == "fp32", payload
    dist.all_reduce(flat_grads, group=group)
    return 1.0 / world


# measurement record of the training step's gradient exchange (bench.py / tools/train_bench.py reset and read it): steps finished, segment
# collectives issued, payload bytes one rank handed to the collective library, host seconds inside segment_ready + finish
GRAD_STATS = {"steps": 0, "segments": 0, "bytes": 0, "host_s": 0.0}
import os as _os

COMM_STREAM = _os.environ.get("A3D_GRAD_COMM_STREAM", "side")


class GradientExchange:
    """The data-parallel gradient exchange in SEGMENTS, each issued the moment the backward pass has finished writing it -- what
    DistributedDataParallel's bucketed all-reduce does behind the reference's training loop (tools/train_net.py:96,110-117: buckets are
    reduced while autograd is still walking the earlier layers).  `allreduce_gradients` is the monolithic form (one collective behind
    the whole backward pass, fully exposed); this is the form the trainer's step uses.

    segments: (lo, hi) element ranges of the flat gradient buffer in the order the backward pass COMPLETES them (box head, res5 + FPN +
    RPN head, res4, res3 for the detector: the flat buffer is laid out in forward order, so the ranges run back to front).  Every
    boundary is a multiple of 4 elements (the casts move 4 per lane).
      begin()                    once per step, before the backward pass
      segment_ready(i, stream)   right after the LAST launch that writes segment i has been enqueued on `stream`: records an event there;
                                 the communication stream (streams.side(2)) waits for it, casts the segment (bf16 payload), hands it to
                                 the collective (async_op: RCCL's own stream runs it) and queues the widening cast behind it -- the host
                                 never waits, the main stream is not touched
      finish() -> factor         the current stream waits for the communication stream; returns what the optimiser multiplies the
                                 gradient by (1 / world for the fp32 payload, 1.0 for bf16: divided before rounding, as in
                                 allreduce_gradients)
    Element-wise the arithmetic is that of allreduce_gradients (a sum all-reduce does not mix elements): at world 2 the parameters after
    any number of steps are bit-identical to the monolithic form's (tests/test_distributed.py, tests/test_gpu_training.py).
    Test rigs: host tensors go through gloo directly; device tensors on a gloo group (several ranks on one GPU) are staged through the
    host per segment -- at segment_ready time, so a segment announced too early still shows up as a wrong bit.
    force=True runs the segments even at world 1 (measurement: every launch, event and collective of the N > 1 step on a 1-GPU box)."""

    def __init__(self, flat_grads: torch.Tensor, segments, group=None, payload: str = "fp32", force: bool = False, widen: bool = True):
        assert payload in ("fp32", "bf16"), payload
        n = flat_grads.numel()
        segs = [(int(lo), int(hi)) for lo, hi in segments]
        assert sorted(segs)[0][0] == 0 and sorted(segs)[-1][1] == n and all(a[1] == b[0] for a, b in zip(sorted(segs), sorted(segs)[1:])), \
            f"segments must tile [0, {n}): {segs}"
        assert all(lo % 4 == 0 and (hi - lo) % 4 == 0 for lo, hi in segs), segs  # (the casts move 4 elements per lane)
        self.flat, self.segments, self.group, self.payload = flat_grads, segs, group, payload
        # widen=False (bf16 payload on the RCCL path only): the reduced gradient stays in the bf16 buffer the collective summed in --
        # `reduced_bf16` -- for an optimiser kernel that reads it there (a3d_sgd_momentum_bf16g: the same update bit for bit, minus one
        # launch per segment and one pass over the flat buffer); flat_grads then keeps this rank's own gradient
        self.widen = bool(widen)
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.active = on and (self.world > 1 or force)
        self._g16 = None
        self._pending: list = []
        self._issued: set = set()
        self._open = False
        self._events: dict = {}
        self._used_streams: list = []

    # ---- one step
    def begin(self):
        for fin in self._pending:  # (a step whose optimiser update never came: complete its collectives, every rank has issued them)
            fin()
        self._pending, self._issued, self._open, self._used_streams = [], set(), True, []

    def segment_ready(self, i: int, stream=None):
        if not (self.active and self._open):
            return
        import time as _time

        t0 = _time.perf_counter()
        assert i not in self._issued, f"segment {i} announced twice"
        self._issued.add(i)
        lo, hi = self.segments[i]
        seg = self.flat[lo:hi]
        GRAD_STATS["segments"] += 1
        GRAD_STATS["bytes"] += (hi - lo) * (2 if self.payload == "bf16" else 4)
        if not seg.is_cuda:
            self._pending.append(self._host_segment(seg))
        elif dist.get_backend(self.group) == "gloo":
            ev = self._event(i)
            ev.record(stream or torch.cuda.current_stream())
            ev.synchronize()  # (test rig: the snapshot is taken NOW -- what an early announcement would get wrong)
            host = seg.cpu()
            done = self._host_segment(host)
            self._pending.append(lambda: (done(), seg.copy_(host)))
        else:
            self._device_segment(i, seg, lo, hi, stream or torch.cuda.current_stream())
        GRAD_STATS["host_s"] += _time.perf_counter() - t0

    def finish(self) -> float:
        if not (self.active and self._open):
            return allreduce_gradients(self.flat, self.group, payload=self.payload)
        import time as _time

        t0 = _time.perf_counter()
        missing = set(range(len(self.segments))) - self._issued
        assert not missing, f"GradientExchange.finish(): segments {sorted(missing)} were never announced"
        for fin in self._pending:
            fin()
        self._pending, self._open = [], False
        if self.flat.is_cuda and dist.get_backend(self.group) != "gloo":
            for st in self._used_streams:
                torch.cuda.current_stream().wait_stream(st)
        GRAD_STATS["steps"] += 1
        GRAD_STATS["host_s"] += _time.perf_counter() - t0
        return 1.0 if self.payload == "bf16" else 1.0 / self.world

    @property
    def reduced_bf16(self):
        """The all-reduced gradient as bf16 when the last finished step left it there (widen=False, bf16 payload, RCCL path), else None."""
        if self.widen or self.payload != "bf16" or self._g16 is None or not self.flat.is_cuda or dist.get_backend(self.group) == "gloo":
            return None
        return self._g16

    # ---- forms
    def _event(self, i):
        ev = self._events.get(i)
        if ev is None:
            ev = self._events[i] = torch.cuda.Event()
        return ev

    def _comm(self):
        from .streams import side

        return side(2, self.flat.device)

    def _comm_for(self, producer):
        """The stream the casts and the collective's enqueue of a segment go to.  "side" (default): the package's communication stream --
        the producer (the weight-gradient stream) is never held up by a transfer.  "producer" (measurement knob): the producer itself."""
        return producer if COMM_STREAM == "producer" else self._comm()

    def _host_segment(self, seg: torch.Tensor):
        """Host tensor `seg` (a view of the flat buffer, or a staged copy): the collective starts now, the returned thunk completes it."""
        if self.payload == "bf16":
            g16 = (seg * (1.0 / self.world)).to(torch.bfloat16)
            w = dist.all_reduce(g16, group=self.group, async_op=True)
            return lambda: (w.wait(), seg.copy_(g16.to(torch.float32)))
        w = dist.all_reduce(seg, group=self.group, async_op=True)
        return w.wait

    def _device_segment(self, i, seg, lo, hi, stream):
        from . import _lib

        comm = self._comm_for(stream)
        ev = self._event(i)
        ev.record(stream)
        if comm not in self._used_streams:
            self._used_streams.append(comm)
        cur = torch.cuda.current_stream()
        torch.cuda.set_stream(comm)  # (the collective orders itself behind the CURRENT stream of the call)
        try:
            comm.wait_event(ev)
            if self.payload == "bf16":
                if self._g16 is None:
                    self._g16 = torch.empty(self.flat.numel(), device=self.flat.device, dtype=torch.bfloat16)
                g16 = self._g16[lo:hi]
                st = comm.cuda_stream
                _lib.check(_lib.lib().a3d_f32_to_bf16_scaled(seg.data_ptr(), g16.data_ptr(), hi - lo, 1.0 / self.world, st), "a3d_f32_to_bf16_scaled")
                dist.all_reduce(g16, group=self.group, async_op=True).wait()  # (the communication stream waits; the host does not)
                if self.widen:
                    _lib.check(_lib.lib().a3d_bf16_to_f32(g16.data_ptr(), seg.data_ptr(), hi - lo, st), "a3d_bf16_to_f32")
            else:
                dist.all_reduce(seg, group=self.group, async_op=True).wait()
        finally:
            torch.cuda.set_stream(cur)
