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
    """Result of gather_records_async.  `wait()` makes the current stream wait for the collective and returns
    (all_records, all_counts); until then the exchange runs on RCCL's own stream, under the next batch's kernels."""

    def __init__(self, all_rec, all_cnt, works, keep):
        self._out, self._works, self._keep = (all_rec, all_cnt), works, keep

    def wait(self):
        for w in self._works:
            w.wait()
        self._works, self._keep = [], None
        return self._out


_blocks_verified = False


def gather_records_async(records: torch.Tensor, rec_count: torch.Tensor, rows: Optional[int] = None) -> GatherHandle:
    """Starts the all-gather of one batch's detection records without blocking the launch stream: the detector's next
    batch is enqueued while the records travel over xGMI (one collective per batch, overlapped with compute).

    `all_gather_into_tensor` needs the SAME block size on every rank, while `shard_range` hands out uneven blocks
    whenever F % world != 0 (10 frames on 4 ranks: 3,3,3,1).  Pass `rows` = ceil(F / world): the block is zero-padded to
    that many frames (count 0) before it travels.  Without `rows` the caller guarantees equal blocks; the first call of a
    process verifies that with one small synchronous all-gather of the block sizes and raises on a mismatch (instead of
    hanging in RCCL); later calls trust the same batching."""
    global _blocks_verified
    if not (dist.is_available() and dist.is_initialized()):
        return GatherHandle(records, rec_count, [], None)
    G = dist.get_world_size()
    rec, cnt = records.contiguous(), rec_count.contiguous()
    back = None
    if rec.is_cuda and dist.get_backend() == "gloo":  # test rigs (several ranks on one GPU): gloo moves host memory
        back = rec.device
        rec, cnt = rec.cpu(), cnt.cpu()
    if rows is not None:
        if rec.shape[0] > rows:
            raise ValueError(f"block of {rec.shape[0]} frames does not fit rows={rows}")
        if rec.shape[0] < rows:
            pad = rows - rec.shape[0]
            rec = torch.cat([rec, rec.new_zeros((pad,) + tuple(rec.shape[1:]))])
            cnt = torch.cat([cnt, cnt.new_zeros((pad,))])
    if not _blocks_verified:
        mine = torch.tensor([rec.shape[0]], device=rec.device, dtype=torch.int64)
        sizes = torch.empty((G,), device=rec.device, dtype=torch.int64)
        dist.all_gather_into_tensor(sizes, mine)
        sizes = sizes.tolist()
        if len(set(sizes)) != 1:
            raise ValueError(f"gather_records: ranks hold blocks of {sizes} frames; pass rows=ceil(F/world) to pad them")
        _blocks_verified = True
    all_rec = torch.empty((G * rec.shape[0],) + tuple(rec.shape[1:]), device=rec.device, dtype=rec.dtype)
    all_cnt = torch.empty((G * cnt.shape[0],), device=cnt.device, dtype=cnt.dtype)
    if back is not None:
        dist.all_gather_into_tensor(all_cnt, cnt)
        dist.all_gather_into_tensor(all_rec, rec)
        return GatherHandle(all_rec.to(back), all_cnt.to(back), [], None)
    w1 = dist.all_gather_into_tensor(all_cnt, cnt, async_op=True)
    w2 = dist.all_gather_into_tensor(all_rec, rec, async_op=True)
    return GatherHandle(all_rec, all_cnt, [w1, w2], (rec, cnt))  # inputs stay referenced until the collective is waited for


def allreduce_gradients(flat_grads: torch.Tensor, group=None) -> float:
    """Data-parallel training (SURVEY.md 8e "training", 8f-1): ONE sum all-reduce of the trainer's flat gradient buffer
    (replaces DistributedDataParallel's bucketed all-reduce behind tools/train_net.py:110).  Returns the factor the
    optimiser applies to turn the sum into DDP's mean (1 / world size); 1.0 when not distributed."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world == 1:
        return 1.0
    dist.all_reduce(flat_grads, group=group)
    return 1.0 / world
