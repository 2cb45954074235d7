"""CPU suite, part 3: the N>1 path (frame sharding + all-gather of detection records) on gloo, world size 2."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from articulation3d_amd.parallel import gather_records, shard_range

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    F_, R, REC = 10, 4, 798
    lo, hi = shard_range(F_, rank, world)
    B = hi - lo
    rec = torch.zeros(B, R, REC)
    cnt = torch.zeros(B, dtype=torch.int32)
    for i, f in enumerate(range(lo, hi)):  # frame f carries f % 3 detections whose score encodes (frame, slot)
        cnt[i] = f % 3
        for r in range(int(cnt[i])):
            rec[i, r, 4] = f + 0.1 * r
    all_rec, all_cnt = gather_records(rec, cnt)
    ok = all_rec.shape == (F_, R, REC) and all_cnt.tolist() == [f % 3 for f in range(F_)]
    for f in range(F_):  # temporal order restored by rank order
        for r in range(f % 3):
            ok = ok and abs(float(all_rec[f, r, 4]) - (f + 0.1 * r)) < 1e-6
    # two batches in flight (what bench.py does: the gather of batch i overlaps batch i+1), waited for in order
    from articulation3d_amd.parallel import gather_records_async

    h1, h2 = gather_records_async(rec, cnt), gather_records_async(rec * 2, cnt)
    (r1, c1), (r2, c2) = h1.wait(), h2.wait()
    ok = ok and torch.equal(r1, all_rec) and torch.equal(r2, all_rec * 2) and torch.equal(c1, all_cnt) and torch.equal(c2, all_cnt)
    # what travels is the LIVE records only: [sum of counts, F], frame-major, in rank order
    rows, cc = gather_records_async(rec, cnt).wait_compact()
    want = [f + 0.1 * r for f in range(F_) for r in range(f % 3)]
    ok = ok and rows.shape == (len(want), REC) and torch.equal(cc, all_cnt) and all(abs(float(a) - b) < 1e-6 for a, b in zip(rows[:, 4], want))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from articulation3d_amd.parallel import shard_range

    for F_, G in ((1024, 8), (10, 4), (3, 8), (64, 1)):
        blocks = [shard_range(F_, r, G) for r in range(G)]
        assert blocks[0][0] == 0 and blocks[-1][1] == F_
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(G - 1))
    assert shard_range(1024, 3, 8) == (384, 512)


def test_gather_records_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _uneven_worker(rank, world, port, q):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from articulation3d_amd.parallel import gather_records, shard_range

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    F_, R, REC = 10, 2, 798  # 10 frames on 4 ranks: blocks of 3, 3, 3, 1 (pipeline.detect_clip's padding path)
    per = (F_ + world - 1) // world
    lo, hi = shard_range(F_, rank, world)
    rec = torch.zeros(hi - lo, R, REC)
    cnt = torch.ones(hi - lo, dtype=torch.int32)
    for i, f in enumerate(range(lo, hi)):
        rec[i, 0, 4] = float(f)
    ok = True
    try:  # without `rows` the uneven blocks are refused with an error on every rank instead of a hang
        gather_records(rec, cnt)
        ok = False
    except ValueError as e:
        ok = "rows=" in str(e)
    try:  # a block that does not fit `rows`: every rank learns it from the counts exchange and ALL raise (nobody hangs) --
        gather_records(rec, cnt, rows=per - 1)  # rank 3's own block (1 frame) fits, it must raise too
        ok = False
    except ValueError as e:
        ok = ok and "rows=" in str(e)
    all_rec, all_cnt = gather_records(rec, cnt, rows=per)
    ok = ok and all_rec.shape == (world * per, R, REC) and all_cnt.shape == (world * per,)
    slots = [r * per + i for r in range(world) for i in range(shard_range(F_, r, world)[1] - shard_range(F_, r, world)[0])]
    ok = ok and [float(all_rec[s, 0, 4]) for s in slots] == [float(f) for f in range(F_)] and all_cnt[slots].tolist() == [1] * F_
    ok = ok and int(all_cnt.sum()) == F_  # padded slots carry count 0
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_records_uneven_shards_gloo_world4():
    """F % world != 0 (10 frames / 4 ranks -> 3,3,3,1) must be padded, or refused loudly."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uneven_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(r, True) for r in range(4)]


def test_gather_records_single_process_is_identity():
    from articulation3d_amd.parallel import gather_records

    from articulation3d_amd.parallel import gather_records_async, scatter_compact

    a, b = torch.zeros(2, 3, 798), torch.zeros(2, dtype=torch.int32)
    x, y = gather_records(a, b)
    assert x is a and y is b
    a = torch.arange(2 * 3 * 798, dtype=torch.float32).view(2, 3, 798)
    b = torch.tensor([2, 1], dtype=torch.int32)
    rows, cnt = gather_records_async(a, b).wait_compact()
    assert rows.shape == (3, 798) and torch.equal(rows[0], a[0, 0]) and torch.equal(rows[1], a[0, 1]) and torch.equal(rows[2], a[1, 0])
    dense = scatter_compact(rows, cnt, 3)
    assert torch.equal(dense[0, :2], a[0, :2]) and torch.equal(dense[1, :1], a[1, :1]) and float(dense[0, 2].abs().sum() + dense[1, 1:].abs().sum()) == 0


def _grad_worker(rank, world, port, q):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from articulation3d_amd.parallel import allreduce_gradients

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)  # rank r holds (r+1) * base
    scale = allreduce_gradients(g)
    want = torch.arange(1000, dtype=torch.float32) * sum(range(1, world + 1))
    ok = bool(torch.equal(g, want)) and abs(scale - 1.0 / world) < 1e-12
    # bf16 payload (configs[4]; DDP's bf16_compress_hook): mean of the bf16-rounded, pre-divided gradients, widened back; factor 1
    g = (torch.arange(1000, dtype=torch.float32) * 0.37 + 0.123) * (rank + 1)
    scale = allreduce_gradients(g, payload="bf16")
    parts = [((torch.arange(1000, dtype=torch.float32) * 0.37 + 0.123) * (r + 1) * (1.0 / world)).to(torch.bfloat16) for r in range(world)]
    want16 = parts[0]
    for p_ in parts[1:]:
        want16 = want16 + p_  # (bf16 sum, as the collective forms it)
    ok = ok and scale == 1.0 and bool(torch.equal(g, want16.to(torch.float32)))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_gloo_world2():
    """The training step's only collective: sum of the flat gradient buffer, mean applied inside the SGD kernel."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _xchg_worker(rank, world, port, q):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from articulation3d_amd.parallel import GRAD_STATS, GradientExchange, allreduce_gradients

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 4096 + 512
    cuts = [0, 512, 1544, 3000, 4096, n]  # five uneven segments, completed back to front as the backward pass does
    segs = [(cuts[i], cuts[i + 1]) for i in (4, 3, 2, 1, 0)]
    ok = True
    for payload in ("fp32", "bf16"):
        gen = torch.Generator().manual_seed(100 + rank)
        p_seg, p_mono = torch.zeros(n), torch.zeros(n)
        g_seg = torch.zeros(n)
        ex = GradientExchange(g_seg, segs, payload=payload)
        assert ex.active and ex.world == world
        for step in range(3):  # three "optimiser steps": parameters must stay bit-identical to the monolithic form's
            g = torch.randn(n, generator=gen) * (1.0 + step)
            g_mono = g.clone()
            f_mono = allreduce_gradients(g_mono, payload=payload)
            ex.begin()
            for i, (lo, hi) in enumerate(segs):  # a segment is written, then announced; later segments are still "in the backward pass"
                g_seg[lo:hi] = g[lo:hi]
                ex.segment_ready(i)
            f_seg = ex.finish()
            ok = ok and f_seg == f_mono and bool(torch.equal(g_seg, g_mono))
            p_seg -= 0.1 * f_seg * g_seg
            p_mono -= 0.1 * f_mono * g_mono
        ok = ok and bool(torch.equal(p_seg, p_mono)) and float(p_seg.abs().sum()) > 0
    ok = ok and GRAD_STATS["segments"] == 2 * 3 * 5 and GRAD_STATS["steps"] == 6
    # a segment that was never announced is an error on the rank that forgot it, before anything is applied
    ex = GradientExchange(torch.zeros(n), segs)
    ex.begin()
    for i in range(4):
        ex.segment_ready(i)
    try:
        ex.finish()
        ok = False
    except AssertionError:
        ex.segment_ready(4)  # (complete the collective the other rank is in as well)
        ex.finish()
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_exchange_in_segments_equals_the_monolithic_allreduce_gloo_world2():
    """the training step's gradient exchange leaves in segments while the backward pass is still running
    (parallel.GradientExchange; DDP's bucketed all-reduce behind tools/train_net.py:96,110-117).  On two gloo ranks, fp32 and bf16
    payloads: gradients and the parameters after three steps are bit-identical to the one-collective form's."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_xchg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_gradient_exchange_single_process_is_inactive():
    from articulation3d_amd.parallel import GradientExchange

    g = torch.arange(16, dtype=torch.float32)
    ex = GradientExchange(g, [(8, 16), (0, 8)])
    assert not ex.active
    ex.begin()
    ex.segment_ready(0)
    assert ex.finish() == 1.0 and torch.equal(g, torch.arange(16, dtype=torch.float32))
    with pytest.raises(AssertionError):
        GradientExchange(g, [(0, 8), (12, 16)])  # segments must tile the buffer


def test_gradient_allreduce_single_process():
    from articulation3d_amd.parallel import allreduce_gradients

    g = torch.ones(8)
    assert allreduce_gradients(g) == 1.0 and torch.equal(g, torch.ones(8))


# ---- BASELINE configs[3] at its real shapes on 8 gloo ranks (no 8-GPU node exists here; the path is covered on CPU) ---
def _clip1024_worker(rank, world, port, q):
    import sys

    import psutil

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from articulation3d_amd.parallel import gather_records, gather_records_async, shard_range
    from articulation3d_amd.pipeline import instances_from_compact

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    F_, R, REC, BATCH = 1024, 100, 798, 32
    per = (F_ + world - 1) // world
    lo, hi = shard_range(F_, rank, world)
    ok = (lo, hi) == (rank * 128, rank * 128 + 128) and per == 128
    proc = psutil.Process()

    def block(det_of_frame):
        """This rank's [128, 100, 798] slot block, filled batch by batch as pipeline.detect_clip fills it (4 batches of 32 frames);
        detection r of clip frame f carries score f + r / 1000 in field 4 and class r % 3 in field 5."""
        rec = torch.zeros(per, R, REC)
        cnt = torch.zeros(per, dtype=torch.int32)
        for s in range(lo, hi, BATCH):
            for f in range(s, min(s + BATCH, hi)):
                n = det_of_frame(f)
                cnt[f - lo] = n
                rec[f - lo, :n, 4] = f + torch.arange(n, dtype=torch.float32) / 1000.0
                rec[f - lo, :n, 5] = (torch.arange(n) % 3).float()
        return rec, cnt

    slots = [r * per + i for r in range(world) for i in range(128)]
    # (a) the realistic operating point, D ~ 4 per frame: ONE gather of the live records for the rank's whole block
    d4 = lambda f: (f * 7) % 9  # 0..8 detections, mean 4
    rec, cnt = block(d4)
    rss0 = proc.memory_info().rss
    rows, all_cnt = gather_records_async(rec, cnt, rows=per).wait_compact()
    grew = proc.memory_info().rss - rss0
    live = sum(d4(f) for f in range(F_))
    ok = ok and rows.shape == (live, REC) and all_cnt.tolist() == [d4(f) for f in range(F_)]
    # what arrived is the live records (13 MB), not the 327 MB slot layout of the clip: the receiving side never builds the dense form
    ok = ok and rows.numel() * 4 == live * REC * 4 and grew < 150 * 2 ** 20
    preds = instances_from_compact(rows, all_cnt, slots, (480, 640), conf_threshold=-1.0, with_masks=False)
    ok = ok and len(preds) == F_
    for f in (0, 1, 127, 128, 129, 511, 512, 1023):  # temporal order across rank boundaries, frame by frame
        want = [f + r / 1000.0 for r in range(d4(f))]
        ok = ok and len(preds[f].scores) == len(want) and all(abs(float(a) - b) < 1e-3 for a, b in zip(preds[f].scores, want))
    ok = ok and [len(p.scores) for p in preds] == [d4(f) for f in range(F_)]
    # (b) per-batch gathers as bench.py issues them: 4 batches of 32 frames, the gather of batch i in flight while batch i + 1 is built
    pending, got = [], []
    for b in range(4):
        sl = slice(b * BATCH, (b + 1) * BATCH)
        pending.append(gather_records_async(rec[sl].contiguous(), cnt[sl].contiguous(), rows=BATCH))
        if len(pending) > 1:
            got.append(pending.pop(0).wait_compact())
    got.append(pending.pop(0).wait_compact())
    for b, (rws, cc) in enumerate(got):  # batch b of every rank, rank-major: clip frames r * 128 + b * 32 + i
        frames = [r * 128 + b * BATCH + i for r in range(world) for i in range(BATCH)]
        ok = ok and cc.tolist() == [d4(f) for f in frames] and rws.shape[0] == sum(d4(f) for f in frames)
        first = [f for f in frames if d4(f)][:1]
        ok = ok and (not first or abs(float(rws[0, 4]) - first[0]) < 1e-3)
    # (c) the stress point D = 100 (SCORE_THRESH_TEST 0: every slot live): the compact form degenerates to the dense one and must still
    # arrive whole and in order -- one 32-frame batch per rank (82 MB gathered on every rank)
    rec100, cnt100 = block(lambda f: 100)
    rws, cc = gather_records_async(rec100[:BATCH].contiguous(), cnt100[:BATCH].contiguous(), rows=BATCH).wait_compact()
    ok = ok and rws.shape == (world * BATCH * 100, REC) and cc.tolist() == [100] * (world * BATCH)
    ok = ok and abs(float(rws[(3 * BATCH + 5) * 100 + 42, 4]) - (3 * 128 + 5 + 0.042)) < 1e-3
    # (d) a block that does not fit `rows` on ONE rank: all eight raise (nobody is left inside the collective)
    try:
        n_mine = BATCH + 1 if rank == 5 else BATCH
        gather_records(rec[:n_mine].contiguous(), cnt[:n_mine].contiguous(), rows=BATCH)
        ok = False
    except ValueError as e:
        ok = ok and "rows=" in str(e)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_clip_of_1024_frames_on_8_ranks_gloo():
    """BASELINE configs[3] at its real shapes: 1024 frames -> 8 contiguous blocks of 128, filled in 4 batches of 32; records of
    100 slots x 798 floats; D ~ 4 and D = 100.  Checks the temporal order of the reassembled clip, that the receiving side holds
    the live records only, bench.py's one-gather-per-batch pipelining, and the all-ranks-raise path at world size 8
    (reference analogue: tools/train_net.py:110-117 launch, evaluation/arti_evaluation.py:195-199 gather)."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_clip1024_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    assert res == [(r, True) for r in range(world)]
