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

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_range(num_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of rank `rank`; concatenation in rank order restores temporal order."""
    per = (num_frames + world - 1) // world
    lo = min(rank * per, num_frames)
    return lo, min(lo + per, num_frames)


def gather_records(records: torch.Tensor, rec_count: torch.Tensor):
    """records [B,R,F] fp32, rec_count [B] int32 (this rank's block) -> ([G*B,R,F], [G*B]) on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return records, rec_count
    G = dist.get_world_size()
    all_rec = torch.empty((G * records.shape[0],) + tuple(records.shape[1:]), device=records.device, dtype=records.dtype)
    all_cnt = torch.empty((G * rec_count.shape[0],), device=rec_count.device, dtype=rec_count.dtype)
    dist.all_gather_into_tensor(all_cnt, rec_count.contiguous())
    dist.all_gather_into_tensor(all_rec, records.contiguous())
    return all_rec, all_cnt


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
