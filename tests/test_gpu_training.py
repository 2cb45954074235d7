"""GPU suite (-m gpu) for the training step: every backward / loss / optimiser kernel through the
C ABI against torch autograd on the CPU (floating-point kernels) or the CPU oracle (oracle/train_oracle.py), then the
whole step against the oracle's loss_and_grads on identical sampled index sets.

Tolerances (fp32): gradients are long sums, compared by relative L2 error per tensor (<= 2e-4: the oracle's own fp32
summation-order noise against float64 is ~1e-5..1e-4 on the deepest layers); matcher labels / indices bit-exact.
"""
import math

import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def l2rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.fixture(scope="module")
def T():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from articulation3d_amd import train_ops

    return train_ops


@pytest.fixture(scope="module")
def ops():
    from articulation3d_amd import ops as o

    return o


WG_CASES = [
    dict(B=2, H=30, W=40, Cin=256, Cout=128, k=1, s=1, p=0),
    dict(B=2, H=30, W=40, Cin=256, Cout=512, k=1, s=2, p=0),
    dict(B=2, H=15, W=20, Cin=128, Cout=128, k=3, s=1, p=1),
    dict(B=1, H=9, W=7, Cin=36, Cout=20, k=3, s=1, p=1),  # ragged channel tiles
    dict(B=1000, H=1, W=1, Cin=1024, Cout=32, k=1, s=1, p=0),  # predictor-shaped linear
    dict(B=3, H=8, W=10, Cin=256, Cout=32, k=1, s=1, p=0),
]


@pytest.mark.parametrize("c", WG_CASES, ids=lambda c: f"{c['Cin']}to{c['Cout']}k{c['k']}s{c['s']}")
def test_conv_wgrad_vs_autograd(T, c):
    torch.manual_seed(1)
    x = torch.randn(c["B"], c["Cin"], c["H"], c["W"])
    w = torch.randn(c["Cout"], c["Cin"], c["k"], c["k"], requires_grad=True)
    y = F.conv2d(x, w, None, c["s"], c["p"])
    dy = torch.randn_like(y)
    y.backward(dy)
    scale = torch.rand(c["Cout"]) + 0.5
    ref = w.grad.permute(0, 2, 3, 1).reshape(c["Cout"], -1)
    dw = torch.full((c["Cout"], c["k"] * c["k"] * c["Cin"]), 7.0, device="cuda")
    T.conv_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), dw, KH=c["k"], KW=c["k"], stride=c["s"], pad=c["p"])
    assert l2rel(dw, ref) < 1e-5
    # folded-BN scale + accumulation + an explicit slice count (deterministic: two runs agree bit for bit)
    dw2 = dw.clone()
    T.conv_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), dw2, KH=c["k"], KW=c["k"], stride=c["s"], pad=c["p"], scale=scale.cuda(),
                 accumulate=True, splitk=3)
    assert l2rel(dw2, ref + ref * scale[:, None]) < 1e-5
    dw3 = dw.clone()
    T.conv_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), dw3, KH=c["k"], KW=c["k"], stride=c["s"], pad=c["p"], scale=scale.cuda(),
                 accumulate=True, splitk=3)
    assert torch.equal(dw2, dw3)


@pytest.mark.parametrize("k,s,wino", [(3, 1, False), (3, 1, True), (1, 1, False), (1, 2, False)])
def test_conv_dgrad_through_forward_kernel(T, ops, k, s, wino):
    """dx = conv(dy, transposed + flipped filter) with the ReLU gate and the residual add fused, vs autograd."""
    torch.manual_seed(2)
    B, Cin, Cout, H, W = 2, 128, 256, 30, 40
    x = torch.randn(B, Cin, H, W, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k) / (k * k * Cin) ** 0.5
    scale = torch.rand(Cout) + 0.5
    pre = torch.randn(B, Cin, H, W)  # the forward tensor whose ReLU produced x: gate = pre > 0
    y = F.conv2d(F.relu(pre) * 0 + x * (pre > 0), w, None, s, k // 2) * scale[None, :, None, None]
    dy = torch.randn_like(y)
    extra = torch.randn(B, Cin, H, W)  # gradient arriving on the identity shortcut
    y.backward(dy)
    ref = x.grad + extra * (pre > 0)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().cuda()
    wt = torch.empty(Cin * k * k * Cout, device="cuda")
    T.weight_transpose(wp, wt, Cout, k, k, Cin, scale=scale.cuda())
    pk = ops.PackedConv(wt.view(Cin, k * k * Cout), None, None, k, k, 1, k // 2, Cout, Cin, k * k * Cout)
    if wino:
        U = torch.empty(16 * Cin * Cout, device="cuda")
        T.wino_weight_transform(wt, U, Cin, Cout)
        pk.w_wino = U.view(16, Cin, Cout)
    dyd = nhwc(dy).cuda()
    gate = nhwc(pre).cuda()
    if s == 1:
        if wino:  # the Winograd form has no residual operand: add the shortcut gradient before gating is not possible -> gate only
            dx = ops.conv2d(dyd, pk, wino=True, gate=gate)
            ref = x.grad
        else:
            dx = ops.conv2d(dyd, pk, res=nhwc(extra).cuda(), gate=gate, wino=False)
    else:
        low = ops.conv2d(dyd, pk, wino=False)
        up = T.zero_insert2(low, H, W)
        dx = up * (gate > 0) + nhwc(extra).cuda() * (gate > 0)  # (test-side combination; the step fuses it into the next conv)
    assert l2rel(dx, nhwc(ref)) < (5e-6 if not wino else 2e-5)


def test_wino_weight_transform_matches_host_packing(T, ops):
    torch.manual_seed(3)
    w = torch.randn(64, 48, 3, 3)
    ref = ops.winograd_weights(w)
    wp = w.permute(0, 2, 3, 1).reshape(64, -1).contiguous().cuda()
    U = torch.empty(16 * 64 * 48, device="cuda")
    T.wino_weight_transform(wp, U, 64, 48)
    assert l2rel(U.view(16, 64, 48), ref) < 1e-6


def test_spatial_gradient_plumbing(T):
    torch.manual_seed(4)
    x = torch.randn(2, 8, 10, 64, device="cuda")
    y = T.zero_insert2(x, 15, 20)
    ref = torch.zeros(2, 15, 20, 64, device="cuda")
    ref[:, ::2, ::2] = x
    assert torch.equal(y, ref)
    T.zero_insert2(x, 15, 20, out=y, accumulate=True)
    assert torch.equal(y, 2 * ref)
    big = torch.randn(2, 16, 20, 64, device="cuda")
    acc = torch.randn(2, 8, 10, 64, device="cuda")
    want = acc + big.view(2, 8, 2, 10, 2, 64).sum((2, 4))
    T.sumpool2_add(big, acc)
    assert l2rel(acc, want) < 1e-6
    dy = torch.randn(5000, 264, device="cuda")
    out = torch.ones(264, device="cuda")
    T.colsum(dy, out, accumulate=True)
    assert l2rel(out, 1 + dy.double().sum(0)) < 1e-6


def test_roi_align_backward_vs_oracle(T, oracle):
    from oracle import train_oracle as TO

    torch.manual_seed(5)
    B, C, R = 2, 64, 37
    feats = {n: torch.randn(B, C, 480 // s, 640 // s, requires_grad=True) for n, s in (("p2", 4), ("p3", 8), ("p4", 16), ("p5", 32))}
    g = torch.Generator().manual_seed(5)
    wh = torch.rand(B, R, 2, generator=g) * torch.tensor([500.0, 400.0]) + 4
    xy = torch.rand(B, R, 2, generator=g) * (torch.tensor([640.0, 480.0]) - wh * 0.8)
    boxes = torch.cat([xy, xy + wh], 2)
    boxes[0, 0] = torch.tensor([-20.0, -30.0, 700.0, 520.0])  # outside the image
    boxes[1, 1] = torch.tensor([100.0, 100.0, 100.5, 100.2])  # tiny
    pooled = TO.roi_pool_fpn_diff(feats, [boxes[0], boxes[1]], 7, 0, True)
    dout = torch.randn_like(pooled)
    pooled.backward(dout)
    dfe = [torch.zeros(B, f.shape[2], f.shape[3], C, device="cuda") for f in feats.values()]
    T.roi_align_fpn_backward(dfe, [1 / 4, 1 / 8, 1 / 16, 1 / 32], boxes.cuda(), nhwc(dout).contiguous().cuda(), P=7,
                             sampling_ratio=0, aligned=True)
    for d, f in zip(dfe, feats.values()):
        assert l2rel(d, nhwc(f.grad)) < 1e-5
    # round 4: the default is the tile-gather form (no atomics): fixed summation order -> two runs agree bit for bit, it ADDS to what
    # dfeat holds, and it agrees with the float-atomics form to fp32 rounding; live counts and compact rows as the forward pooler's
    args = ([1 / 4, 1 / 8, 1 / 16, 1 / 32], boxes.cuda(), nhwc(dout).contiguous().cuda())
    again = [torch.ones_like(d) for d in dfe]
    T.roi_align_fpn_backward(again, *args, P=7, sampling_ratio=0, aligned=True)
    scat = [torch.zeros_like(d) for d in dfe]
    T.roi_align_fpn_backward(scat, *args, P=7, sampling_ratio=0, aligned=True, scatter=True)
    for d, a2, sc in zip(dfe, again, scat):
        assert torch.equal(a2 - 1.0, (d + 1.0) - 1.0)
        assert l2rel(d, sc) < 1e-6
    count = torch.tensor([20, 0], dtype=torch.int32, device="cuda")
    off = torch.tensor([0, 20], dtype=torch.int32, device="cuda")
    rows = nhwc(dout).contiguous().cuda()[:20].contiguous()
    part, part_s = [torch.zeros_like(d) for d in dfe], [torch.zeros_like(d) for d in dfe]
    T.roi_align_fpn_backward(part, args[0], args[1], rows, P=7, sampling_ratio=0, aligned=True, count=count, row_offset=off)
    T.roi_align_fpn_backward(part_s, args[0], args[1], rows, P=7, sampling_ratio=0, aligned=True, count=count, row_offset=off, scatter=True)
    for a2, sc in zip(part, part_s):
        assert float(a2[1].abs().max()) == 0.0 and l2rel(a2, sc) < 1e-6


def test_matcher_bit_exact_vs_oracle(T, oracle):
    from oracle import train_oracle as TO

    cfg, tc = oracle.OracleCfg(), TO.TrainCfg()
    feat_hw = [(120, 160), (60, 80), (30, 40), (15, 20), (8, 10)]
    anchors = TO.all_anchors(feat_hw, cfg)
    tg = TO.synthetic_targets(3)
    tg[2] = (tg[2][0][:0], tg[2][1][:0])  # an image without ground truth
    Gmax = 8
    gtb = torch.zeros(3, Gmax, 4)
    cnt = torch.zeros(3, dtype=torch.int32)
    for i, (b, _) in enumerate(tg):
        gtb[i, : len(b)] = b
        cnt[i] = len(b)
    midx, lab = T.match_boxes(anchors.cuda(), gtb.cuda(), cnt.cuda(), thresholds=tc.rpn_iou_thresholds, labels=(0, -1, 1),
                              allow_low_quality=True, shared=True)
    for i, (b, _) in enumerate(tg):
        ri, rl = TO.matcher(TO.pairwise_iou(b, anchors), tc.rpn_iou_thresholds, (0, -1, 1), True)
        assert torch.equal(lab[i].cpu(), rl)
        assert torch.equal(midx[i].cpu().long(), ri)
    # proposal matcher: per-image boxes, one threshold, no low-quality matches
    g = torch.Generator().manual_seed(6)
    props = torch.rand(3, 500, 4, generator=g) * 300
    props[..., 2:] += props[..., :2] + 5
    props[0, :4] = tg[0][0][:4] if len(tg[0][0]) >= 4 else props[0, :4]
    midx, lab, iou = T.match_boxes(props.cuda(), gtb.cuda(), cnt.cuda(), thresholds=(0.5,), labels=(0, 1), allow_low_quality=False,
                                   return_iou=True)
    for i, (b, _) in enumerate(tg):
        q = TO.pairwise_iou(b, props[i])
        ri, rl = TO.matcher(q, (0.5,), (0, 1), False)
        assert torch.equal(lab[i].cpu(), rl) and torch.equal(midx[i].cpu().long(), ri)
        if len(b):
            assert torch.equal(iou[i].cpu(), q.max(0)[0])


def test_rpn_and_box_losses_vs_oracle(T, oracle):
    from oracle import train_oracle as TO

    cfg, tc = oracle.OracleCfg(), TO.TrainCfg()
    feat_hw = [(30, 40), (15, 20), (8, 10)]
    strides = [16, 32, 64]
    sizes = (128, 256, 512)
    torch.manual_seed(7)
    B, A, CH = 2, 3, 32
    anchors = torch.cat([oracle.grid_anchors(h, w, s, z, cfg.anchor_ratios) for (h, w), s, z in zip(feat_hw, strides, sizes)], 0)
    tg = TO.synthetic_targets(B)
    gen = torch.Generator().manual_seed(7)
    matched = TO.match_anchors(anchors, [t[0] for t in tg], tc)
    labels = torch.stack([TO.sample_anchors(m[1], tc, gen) for m in matched])
    heads = [torch.randn(B, h, w, CH) for h, w in feat_hw]
    logits = [h[..., :A].reshape(B, -1).clone().requires_grad_(True) for h in heads]
    deltas = [h[..., A : 5 * A].reshape(B, -1, 4).clone().requires_grad_(True) for h in heads]
    mgt = torch.stack([t[0][m[0]] for t, m in zip(tg, matched)])
    ref = TO.rpn_losses(logits, deltas, anchors, labels, mgt, cfg, tc)
    (ref["loss_rpn_cls"] + ref["loss_rpn_loc"]).backward()
    Gmax = 8
    gtb = torch.zeros(B, Gmax, 4)
    for i, (b, _) in enumerate(tg):
        gtb[i, : len(b)] = b
    midx = torch.stack([m[0] for m in matched]).int()
    cell = torch.stack([oracle.cell_anchors(z, cfg.anchor_ratios) for z in sizes])
    loss, dheads = T.rpn_loss([h.cuda() for h in heads], strides, cell, labels.cuda(), midx.cuda(), gtb.cuda(), A=A,
                              weights=cfg.rpn_weights, normalizer=tc.rpn_batch_per_image * B)
    assert abs(loss[0].item() - ref["loss_rpn_cls"].item()) < 1e-5 * abs(ref["loss_rpn_cls"].item()) + 1e-7
    assert abs(loss[1].item() - ref["loss_rpn_loc"].item()) < 1e-5 * abs(ref["loss_rpn_loc"].item()) + 1e-7
    for l, dh in enumerate(dheads):
        dh = dh.cpu()
        assert l2rel(dh[..., :A].reshape(B, -1), logits[l].grad) < 1e-5
        assert torch.equal(dh[..., A : 5 * A].reshape(B, -1, 4), deltas[l].grad)  # +-1/normalizer or 0
        assert float(dh[..., 5 * A :].abs().max()) == 0.0

    # box losses
    M, K, pitch = 700, cfg.num_classes, 32
    pred = torch.randn(M, pitch)
    cls = torch.randint(0, K + 1, (M,))
    boxes = torch.rand(M, 4) * 300
    boxes[:, 2:] += boxes[:, :2] + 8
    gtbx = boxes + torch.randn(M, 4) * 6
    gtbx[:, 2:] = torch.maximum(gtbx[:, 2:], gtbx[:, :2] + 4)
    sc = pred[:, : K + 1].clone().requires_grad_(True)
    dl = pred[:, K + 1 : K + 1 + 4 * K].clone().requires_grad_(True)
    rb = TO.box_losses(sc, dl, boxes, cls, gtbx, cfg)
    (rb["loss_cls"] + rb["loss_box_reg"]).backward()
    loss, dpred = T.box_loss(pred.cuda(), cls.int().cuda(), boxes.cuda(), gtbx.cuda(), num_classes=K, weights=cfg.box_weights)
    assert abs(loss[0].item() - rb["loss_cls"].item()) < 1e-5 * rb["loss_cls"].item()
    assert abs(loss[1].item() - rb["loss_box_reg"].item()) < 1e-5 * rb["loss_box_reg"].item()
    dpred = dpred.cpu()
    assert l2rel(dpred[:, : K + 1], sc.grad) < 1e-5
    assert torch.equal(dpred[:, K + 1 : K + 1 + 4 * K], dl.grad)
    assert float(dpred[:, K + 1 + 4 * K :].abs().max()) == 0.0


def test_device_samplers_draw_valid_uniform_subsets(T):
    """a3d_sample_labels / a3d_sample_rois: detectron2's subsample_labels rules (counts, classes) hold exactly; the draw is
    reproducible per seed, changes with the seed and is uniform over the candidates."""
    torch.manual_seed(9)
    B, N = 3, 76740
    lab = torch.full((B, N), -1, dtype=torch.int8)
    lab[0, torch.randperm(N)[:40]] = 1      # fewer positives than the cap: all kept
    lab[0, torch.randperm(N)[:60000]] = 0
    lab[1, :500] = 1                        # more positives than the cap
    lab[1, 1000:70000] = 0
    lab[2, 5:105] = 0                       # fewer negatives than requested, no positives
    out = T.sample_labels(lab.cuda(), num=256, max_pos=128, seed=5).cpu()
    for b, (npos, nneg) in enumerate([(int((lab[0] == 1).sum()), 256 - int((lab[0] == 1).sum())), (128, 128), (0, 100)]):
        assert int((out[b] == 1).sum()) == npos and int((out[b] == 0).sum()) == nneg
        assert bool(((out[b] == 1) <= (lab[b] == 1)).all()) and bool(((out[b] == 0) <= (lab[b] == 0)).all())
    assert torch.equal(out, T.sample_labels(lab.cuda(), num=256, max_pos=128, seed=5).cpu())
    assert not torch.equal(out, T.sample_labels(lab.cuda(), num=256, max_pos=128, seed=6).cpu())
    # the draw IS the k smallest hashed keys of each class (csrc/train_sample.hip sample_key, replayed here): the round-4 three-scan kernel
    # and the radix select it replaced pick the same sets
    import numpy as np

    def mix32(x):
        x = x.astype(np.uint32)
        x ^= x >> np.uint32(16)
        x = (x * np.uint32(0x85EBCA6B)).astype(np.uint32)
        x ^= x >> np.uint32(13)
        x = (x * np.uint32(0xC2B2AE35)).astype(np.uint32)
        x ^= x >> np.uint32(16)
        return x

    with np.errstate(over="ignore"):
        idx = np.arange(N, dtype=np.uint32)
        for b in range(B):
            hb = mix32(np.array([np.uint32(5) ^ np.uint32(0) ^ (np.uint32(0x9E3779B9) * np.uint32(b + 1))], dtype=np.uint32))[0]
            h = mix32(hb ^ (np.uint32(0x85EBCA6B) * (idx + np.uint32(1))).astype(np.uint32))
            key = ((h >> np.uint32(1)).astype(np.uint64) << np.uint64(17)) | idx.astype(np.uint64)
            lb = lab[b].numpy()
            npos = min(int((lb == 1).sum()), 128)
            nneg = min(int((lb == 0).sum()), 256 - npos)
            want = np.full(N, -1, dtype=np.int8)
            for cls, k in ((1, npos), (0, nneg)):
                ids = np.nonzero(lb == cls)[0]
                want[ids[np.argsort(key[ids])[:k]]] = cls
            assert np.array_equal(out[b].numpy(), want), b
    # uniformity: over 200 seeds each of the 500 positives of image 1 is kept ~128/500 of the time
    hits = torch.zeros(500)
    for sd in range(200):
        hits += (T.sample_labels(lab[1:2].cuda(), num=256, max_pos=128, seed=100 + sd).cpu()[0, :500] == 1).float()
    p = 128 / 500
    z = (hits - 200 * p) / (200 * p * (1 - p)) ** 0.5
    assert float(z.abs().max()) < 4.5 and abs(float(z.mean())) < 0.3

    # proposals: [boxes | gt] with a matcher result; image 1 has no ground truth
    Bn, R, Gmax, K = 2, 1000, 16, 2
    props = torch.rand(Bn, R, 4) * 300
    props[..., 2:] += props[..., :2] + 4
    pcount = torch.tensor([900, 1000], dtype=torch.int32)
    gtb = torch.rand(Bn, Gmax, 4) * 300
    gtb[..., 2:] += gtb[..., :2] + 30
    gcount = torch.tensor([5, 0], dtype=torch.int32)
    gtc = torch.randint(0, K, (Bn, Gmax), dtype=torch.int32)
    allb, bcount = T.append_gt_boxes(props.cuda(), pcount.cuda(), gtb.cuda(), gcount.cuda())
    assert bcount.tolist() == [905, 1000]
    assert torch.equal(allb[0, :900].cpu(), props[0, :900]) and torch.equal(allb[0, 900:905].cpu(), gtb[0, :5]) and float(allb[0, 905:].abs().max()) == 0
    midx, plab = T.match_boxes(allb, gtb.cuda(), gcount.cuda(), thresholds=(0.5,), labels=(0, 1), allow_low_quality=False, box_count=bcount)
    rb, rg, rcl, ridx, rcnt = T.sample_rois(allb, bcount, gtb.cuda(), gtc.cuda(), gcount.cuda(), midx, plab, num_classes=K, num=512, max_fg=128, seed=3)
    assert rcnt.tolist() == [512, 512]
    for b in range(Bn):
        idx = ridx[b].cpu().long()
        assert len(set(idx.tolist())) == 512 and int(idx.max()) < int(bcount[b])  # no repeats, only live boxes
        cls_all = torch.where(plab[b].cpu() == 1, gtc[b][midx[b].cpu().long()].long(), torch.tensor(K)) if gcount[b] > 0 else torch.full((R + Gmax,), K)
        assert torch.equal(rcl[b].cpu().long(), cls_all[idx])
        assert torch.equal(rb[b].cpu(), allb[b].cpu()[idx])
        nfg = int((cls_all[: int(bcount[b])] < K).sum())
        assert int((rcl[b] < K).sum()) == min(nfg, 128)
        if gcount[b] > 0:
            assert torch.equal(rg[b].cpu(), gtb[b][midx[b].cpu().long()[idx]])
            assert nfg >= 5  # the appended ground-truth boxes match themselves


def test_sgd_momentum_vs_oracle(T, oracle):
    from oracle import train_oracle as TO

    tc = TO.TrainCfg()
    torch.manual_seed(8)
    P = {"w": torch.randn(1000)}
    bufs = {}
    p = P["w"].clone().cuda()
    buf = torch.zeros(1000, device="cuda")
    for it in range(3):
        g = torch.randn(1000)
        lr = TO.lr_at(it, tc)
        TO.sgd_step(P, {"w": g}, bufs, lr, tc)
        T.sgd_momentum(p, (2 * g).cuda(), buf, lr=lr, momentum=tc.momentum, weight_decay=tc.weight_decay, grad_scale=0.5, first=it == 0)
        assert l2rel(p, P["w"]) < 1e-6


# ------------------------------------------------------------------------------------------ the whole step
@pytest.fixture(scope="module")
def trainer_and_ref(hip_model, oracle, oracle_params):
    """One forward/backward of the HIP trainer on 2 synthetic frames + the oracle's loss_and_grads on the same
    sampled index sets and the same proposals (proposal selection is discontinuous -> stage-wise, like the inference suite)."""
    from articulation3d_amd.training import DetectorTrainer
    from oracle import train_oracle as TO

    torch.set_num_threads(min(32, torch.get_num_threads()))
    frames = oracle.synthetic_frames(2)
    tg = TO.synthetic_targets(2)
    tr = DetectorTrainer(hip_model, seed=11)
    p0 = {k: v.cpu() for k, v in tr.export_state_dict().items()}
    losses, aux = tr.forward_backward(torch.from_numpy(frames).cuda(), [t[0] for t in tg], [t[1] for t in tg])
    torch.cuda.synchronize()
    pb, pc = aux["proposals"]
    rc, ri = aux["roi_count"].cpu(), aux["roi_index"].cpu()
    roi_idx = [ri[i, : int(rc[i])].long() for i in range(2)]  # the index sets the device sampler drew
    live_rows = torch.cat([torch.arange(int(rc[i])) + i * ri.shape[1] for i in range(2)])
    samples = dict(anchor_labels=aux["anchor_labels"].cpu(), roi_idx=roi_idx, proposals=[pb[i, : int(pc[i])].cpu() for i in range(2)])
    aux["roi_idx"], aux["live_rows"] = roi_idx, live_rows
    cfg, tc = oracle.OracleCfg(), TO.TrainCfg()
    rl, rg, raux = TO.loss_and_grads(oracle.frames_to_chw(frames), tg, oracle_params, cfg, tc, samples=samples)
    # Second oracle evaluation with the ReLU gates of the HIP forward pass.  A ReLU's derivative is discontinuous in its
    # input: the two fp32 forward passes differ by ~1e-5 relative, which flips ~1e-5 of the gates per layer and moves the
    # deep-layer gradients by ~sqrt(flipped fraction) ~ 1e-3 per layer, 2e-2 accumulated at res3 (measured; the oracle's
    # own fp32-vs-fp64 distance is the same size).  Like every discontinuous stage of this suite the backward pass is
    # therefore ALSO compared on identical gates, where only summation-order noise remains.
    gates = [(g.permute(0, 3, 1, 2) if g.dim() == 4 else g[live_rows.to(g.device)]).cpu() > 0 for g in aux["relu_outputs"]]
    state = {"n": 0}
    real = torch.nn.functional

    class _F:  # torch.nn.functional with relu / relu_ replaced by "multiply by the given gate" after the frozen layers
        def __getattr__(self, name):
            return getattr(real, name)

        def relu(self, x, inplace=False):
            i = state["n"]
            state["n"] += 1
            return real.relu(x) if i < 10 else x * gates[i - 10]  # 10 = stem + 3x3 res2 ReLUs (frozen, not kept)

        relu_ = relu

    old = oracle.F
    oracle.F = _F()
    try:
        _, rg_gated, _ = TO.loss_and_grads(oracle.frames_to_chw(frames), tg, oracle_params, cfg, tc, samples=samples)
    finally:
        oracle.F = old
    assert state["n"] == 10 + len(gates), (state["n"], len(gates))
    return tr, p0, losses, aux, rl, rg, raux, tg, frames, rg_gated


def test_training_params_roundtrip(trainer_and_ref, oracle_params):
    tr, p0 = trainer_and_ref[0], trainer_and_ref[1]
    from oracle import train_oracle as TO

    names = TO.trainable_names(oracle_params)
    assert set(names) == set(p0), set(names) ^ set(p0)
    assert sum(v.numel() for v in p0.values()) == 41077786
    for k in names:
        assert torch.equal(p0[k], oracle_params[k]), k


def test_training_labels_and_losses_match_oracle(trainer_and_ref):
    tr, p0, losses, aux, rl, rg, raux, tg, frames, _ = trainer_and_ref
    # matcher on the anchors: identical to the oracle's (before sub-sampling)
    midx, lab = aux["anchor_match"]
    for i in range(2):
        assert torch.equal(lab[i].cpu(), raux["anchor_match"][i][1])
        assert torch.equal(midx[i].cpu().long(), raux["anchor_match"][i][0])
        n = len(aux["roi_idx"][i])
        assert torch.equal(aux["roi_cls"][i, :n].cpu().long(), raux["roi_cls"][i])  # class labels of the sampled ROIs
    for k in ("loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg"):
        a, b = losses[k].item(), rl[k].item()
        assert abs(a - b) <= 2e-4 * abs(b) + 1e-7, (k, a, b)


def test_training_gradients_match_autograd(trainer_and_ref):
    tr, p0, losses, aux, rl, rg, raux, tg, frames, rg_gated = trainer_and_ref
    g = {k: v.cpu() for k, v in tr.export_grads().items()}
    assert set(g) == set(rg) == set(rg_gated)
    import os
    if os.environ.get("A3D_TRAIN_DEBUG"):
        for k in rg:
            print(f"{l2rel(g[k], rg_gated[k]):10.3e} {l2rel(g[k], rg[k]):10.3e}  {float(rg[k].norm()):10.3e}  {k}")
    worst_gated = max((l2rel(g[k], rg_gated[k]), k) for k in rg)
    worst_free = max((l2rel(g[k], rg[k]), k) for k in rg)
    print("worst relative L2 gradient error: identical gates", worst_gated, " free-running", worst_free)
    for k in rg:  # identical ReLU gates: fp32 summation-order noise only
        assert l2rel(g[k], rg_gated[k]) < 2e-4, (k, l2rel(g[k], rg_gated[k]))
    for k in rg:  # free-running fp32 vs fp32: bounded by the gate-flip noise explained in the fixture
        assert l2rel(g[k], rg[k]) < 5e-2, (k, l2rel(g[k], rg[k]))


def test_training_sgd_update_and_loss_decreases(trainer_and_ref, oracle, oracle_params):
    from oracle import train_oracle as TO

    tr, p0, losses, aux, rl, rg, raux, tg, frames, _ = trainer_and_ref
    tc = TO.TrainCfg()
    P = {k: v.clone() for k, v in oracle_params.items()}
    g_hip = {k: v.cpu() for k, v in tr.export_grads().items()}
    TO.sgd_step(P, g_hip, {}, TO.lr_at(0, tc), tc)  # the oracle's update applied to the HIP gradients
    tr.optimizer_step()
    p1 = {k: v.cpu() for k, v in tr.export_state_dict().items()}
    for k in g_hip:
        assert l2rel(p1[k], P[k]) < 1e-6, k
    # a few more steps on the same batch at a larger learning rate: the total loss goes down
    tr.s.warmup_iters = 0
    tr.s.base_lr = 0.002  # (0.01 without warm-up is occasionally unstable on this random-init net)
    fr = torch.from_numpy(frames).cuda()
    hist = []
    for _ in range(10):
        l, _ = tr.step(fr, [t[0] for t in tg], [t[1] for t in tg], samples=dict(anchor_labels=aux["anchor_labels"].cpu(), roi_idx=aux["roi_idx"]))
        hist.append(sum(v.item() for v in l.values()))
    assert hist[-1] < hist[0], hist


def test_training_overfits_a_fixed_batch(hip_model, oracle):
    """40 sync-free steps (device-side sampling with a fresh draw every step) on two fixed frames at lr 0.002 (measured:
    1.48 -> 0.21-0.25 in every repeat; at 0.005 and above momentum SGD without warm-up spikes late in some runs): every
    loss term stays finite and the total drops to less than half -- the step trains."""
    from articulation3d_amd.training import DetectorTrainer, SolverCfg
    from oracle import train_oracle as TO

    frames = torch.from_numpy(oracle.synthetic_frames(2)).cuda()
    tg = TO.synthetic_targets(2)
    tr = DetectorTrainer(hip_model, SolverCfg(base_lr=0.002, warmup_iters=0), seed=3)
    hist = []
    for _ in range(40):
        losses, _ = tr.step(frames, [t[0] for t in tg], [t[1] for t in tg])
        hist.append(torch.stack([v for v in losses.values()]))
    hist = torch.stack(hist).cpu()  # one host read at the end
    assert bool(torch.isfinite(hist).all())
    first, last = hist[:5].sum(1).mean().item(), hist[-5:].sum(1).mean().item()
    print("total loss first/last 5 steps:", first, last)
    assert last < 0.5 * first, (first, last)
    sd = tr.export_state_dict()
    assert all(bool(torch.isfinite(v).all()) for v in sd.values())


# ------------------------------------------------------------------------------------------ bf16 autocast arithmetic
def _bf(t):
    return t.bfloat16().double()


@pytest.mark.parametrize("c", [dict(B=2, H=30, W=40, Cin=256, Cout=128, k=1, s=1, p=0), dict(B=2, H=30, W=40, Cin=256, Cout=512, k=1, s=2, p=0),
                               dict(B=2, H=15, W=20, Cin=128, Cout=192, k=3, s=1, p=1), dict(B=700, H=1, W=1, Cin=1024, Cout=32, k=1, s=1, p=0),
                               # (round 4: the transposed-read weight-gradient form -- ragged channel tiles, several chunks and slices, 256-wide ci tile)
                               dict(B=3, H=31, W=39, Cin=160, Cout=136, k=3, s=1, p=1), dict(B=2, H=60, W=80, Cin=256, Cout=256, k=3, s=1, p=1),
                               dict(B=5, H=23, W=17, Cin=320, Cout=72, k=1, s=1, p=0)],
                         ids=lambda c: f"{c['B']}x{c['H']}x{c['Cin']}to{c['Cout']}k{c['k']}s{c['s']}")
def test_bf16_conv_and_wgrad_kernels(T, ops, c):
    """precision=1: operands rounded to bf16 (nearest-even), products accumulated in fp32 -- against float64 convolutions of the
    SAME bf16-rounded operands (what remains is fp32 accumulation noise), with bias / ReLU / residual / gate fused."""
    torch.manual_seed(12)
    x = torch.randn(c["B"], c["Cin"], c["H"], c["W"])
    w = torch.randn(c["Cout"], c["Cin"], c["k"], c["k"]) / (c["k"] * c["k"] * c["Cin"]) ** 0.5
    b = torch.randn(c["Cout"])
    y_ref = F.conv2d(_bf(x), _bf(w), b.double(), c["s"], c["p"])
    r = torch.randn_like(y_ref).float()
    g = torch.randn_like(y_ref).float()
    pk = ops.pack_conv(w, b, None, c["s"], c["p"], ops.ACT_RELU)
    y = ops.conv2d(nhwc(x).cuda(), pk, precision=1, res=nhwc(r).cuda(), gate=nhwc(g).cuda())
    want = F.relu(y_ref + r.double()) * (g > 0)
    assert l2rel(y.permute(0, 3, 1, 2), want) < 2e-6
    y32 = ops.conv2d(nhwc(x).cuda(), pk, res=nhwc(r).cuda(), gate=nhwc(g).cuda(), wino=False)
    assert 1e-4 < l2rel(y, y32) < 2e-2  # it really is bf16 arithmetic, and close to fp32
    # weight gradient
    dy = torch.randn(c["B"], c["Cout"], y_ref.shape[2], y_ref.shape[3])
    wd = _bf(w).clone().requires_grad_(True)
    F.conv2d(_bf(x), wd, None, c["s"], c["p"]).backward(_bf(dy))
    ref = wd.grad.permute(0, 2, 3, 1).reshape(c["Cout"], -1)
    dw = torch.empty((c["Cout"], c["k"] * c["k"] * c["Cin"]), device="cuda")
    T.conv_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), dw, KH=c["k"], KW=c["k"], stride=c["s"], pad=c["p"], precision=1)
    assert l2rel(dw, ref) < 2e-6


def test_bf16_training_step_tracks_fp32(hip_model, oracle):
    """The bf16 step (autocast arithmetic) on the same batch and the same sampled sets: losses within 1 %, gradients within a
    few percent of the fp32 step's (bf16 has an 8-bit mantissa), and it trains (loss drops over 30 steps)."""
    from articulation3d_amd.training import DetectorTrainer, SolverCfg
    from oracle import train_oracle as TO

    frames = torch.from_numpy(oracle.synthetic_frames(2)).cuda()
    tg = TO.synthetic_targets(2)
    gb, gc = [t[0] for t in tg], [t[1] for t in tg]
    t32 = DetectorTrainer(hip_model, seed=5, precision="fp32")
    l32, aux = t32.forward_backward(frames, gb, gc)
    rc, ri = aux["roi_count"].cpu(), aux["roi_index"].cpu()
    samples = dict(anchor_labels=aux["anchor_labels"].cpu(), roi_idx=[ri[i, : int(rc[i])].long() for i in range(2)])
    g32 = {k: v.cpu() for k, v in t32.export_grads().items()}
    t16 = DetectorTrainer(hip_model, seed=5, precision="bf16")
    l16, aux16 = t16.forward_backward(frames, gb, gc, samples=samples)
    g16 = {k: v.cpu() for k, v in t16.export_grads().items()}
    for k in ("loss_rpn_cls", "loss_rpn_loc"):  # (the box losses see different proposals in the two runs: not comparable)
        assert abs(l16[k].item() - l32[k].item()) < 0.02 * abs(l32[k].item()) + 1e-3, (k, l16[k].item(), l32[k].item())
    assert all(math.isfinite(v.item()) for v in l16.values())
    # The proposals of the two runs differ (different head outputs), so the given ROI index sets select different boxes:
    # only the RPN head's gradients (fixed anchors, identical sampled labels) are comparable tensor by tensor.
    rpn = [k for k in g32 if k.startswith("proposal_generator.rpn_head.")]
    errs = {k: l2rel(g16[k], g32[k]) for k in rpn}
    print("bf16 vs fp32 RPN-head gradients, relative L2:", {k.split("rpn_head.")[1]: round(v, 4) for k, v in errs.items()})
    # objectness (smooth BCE) gradients agree to ~1 %; the box-delta loss is L1, whose gradient is the SIGN of (pred - target) and
    # flips for the few positive anchors whose bf16-rounded prediction crosses the target -> tens of percent on those tensors
    assert max(v for k, v in errs.items() if "objectness" in k) < 0.05
    assert max(errs.values()) < 0.5
    assert all(bool(torch.isfinite(v).all()) for v in g16.values())
    for k in g32:  # same scale everywhere
        assert 0.5 < float(g16[k].norm() / (g32[k].norm() + 1e-30)) < 2.0, k
    tr = DetectorTrainer(hip_model, SolverCfg(base_lr=0.002, warmup_iters=0), seed=3, precision="bf16")
    hist = torch.stack([torch.stack(list(tr.step(frames, gb, gc)[0].values())).sum() for _ in range(30)]).cpu()
    assert bool(torch.isfinite(hist).all()) and hist[-5:].mean() < 0.6 * hist[:5].mean(), hist.tolist()


def test_bf16x3_training_step_matches_fp32(hip_model, oracle):
    """DetectorTrainer(precision="bf16x3"): forward / data-gradient launches of the non-Winograd layers on the bf16 pipe with the
    exact 3-way operand split (fp32-grade).  Same batch, same sampled sets: the RPN losses agree with the fp32 step to 1e-5
    relative, the objectness gradients to 1e-4 and the box-delta gradients to 5e-3 (both runs are fp32-accurate; what differs is
    rounding order, amplified where an L1 sign or a ReLU gate flips -- the free-running HIP-vs-oracle comparison of the fp32 step
    allows 5e-2 for the same reason)."""
    from articulation3d_amd.training import DetectorTrainer
    from oracle import train_oracle as TO

    frames = torch.from_numpy(oracle.synthetic_frames(2)).cuda()
    tg = TO.synthetic_targets(2)
    gb, gc = [t[0] for t in tg], [t[1] for t in tg]
    t32 = DetectorTrainer(hip_model, seed=5, precision="fp32")
    l32, aux = t32.forward_backward(frames, gb, gc)
    rc, ri = aux["roi_count"].cpu(), aux["roi_index"].cpu()
    samples = dict(anchor_labels=aux["anchor_labels"].cpu(), roi_idx=[ri[i, : int(rc[i])].long() for i in range(2)])
    g32 = {k: v.cpu() for k, v in t32.export_grads().items()}
    tx3 = DetectorTrainer(hip_model, seed=5, precision="bf16x3")
    lx3, _ = tx3.forward_backward(frames, gb, gc, samples=samples)
    gx3 = {k: v.cpu() for k, v in tx3.export_grads().items()}
    for k in ("loss_rpn_cls", "loss_rpn_loc"):
        assert abs(lx3[k].item() - l32[k].item()) < 1e-5 * abs(l32[k].item()) + 1e-7, (k, lx3[k].item(), l32[k].item())
    rpn = [k for k in g32 if k.startswith("proposal_generator.rpn_head.")]
    errs = {k: l2rel(gx3[k], g32[k]) for k in rpn}
    print("bf16x3 vs fp32 RPN-head gradients, relative L2:", {k.split("rpn_head.")[1]: float("%.2e" % v) for k, v in errs.items()})
    assert max(v for k, v in errs.items() if "objectness" in k) < 1e-4
    assert max(errs.values()) < 5e-3
    assert all(bool(torch.isfinite(v).all()) for v in gx3.values())


def test_bf16x3_weight_gradient_is_fp32_grade(T):
    """a3d_wgrad_desc.precision == 2: the pixel-reduction GEMM with both operands split exactly into three bf16 terms.  Against
    the float64 weight gradient its error is no larger than the fp32 MFMA kernel's (3x3 stride 2 with padding, ragged channel
    tiles, split-K over the pixels)."""
    torch.manual_seed(31)
    B, Cin, Cout, H, W, k, s, p = 3, 96, 136, 37, 41, 3, 2, 1
    x = torch.randn(B, Cin, H, W)
    w = (torch.randn(Cout, Cin, k, k) / (k * k * Cin) ** 0.5).double().requires_grad_(True)
    y = F.conv2d(x.double(), w, None, s, p)
    dy = torch.randn(y.shape)
    y.backward(dy.double())
    ref = w.grad.permute(0, 2, 3, 1).reshape(Cout, -1)
    err = {}
    for prec in (0, 2):
        dw = torch.empty((Cout, k * k * Cin), device="cuda")
        T.conv_wgrad(nhwc(x).cuda(), nhwc(dy).cuda(), dw, KH=k, KW=k, stride=s, pad=p, precision=prec)
        err[prec] = ((dw.double().cpu() - ref).norm() / ref.norm()).item()
    print("weight gradient, relative L2 error vs float64: fp32 MFMA %.3e, bf16x3 %.3e" % (err[0], err[2]))
    assert err[2] < 1e-6 and err[2] <= 1.05 * err[0]


def test_reference_training_loop_is_a_drop_in(oracle, oracle_params):
    """tools/train_net.py:84-104 -> detectron2 SimpleTrainer.run_step on this package (training forward raised):
    `loss_dict = model(data); losses = sum(loss_dict.values()); optimizer.zero_grad(); losses.backward(); optimizer.step()` with
    the step1_bbox configuration.  Checked against DetectorTrainer driven directly: same losses, same parameters after the step,
    and the updated weights reach the inference path when the model goes back to eval mode."""
    from conftest import ROOT
    from articulation3d_amd.config import get_cfg, get_planercnn_cfg_defaults
    from articulation3d_amd.engine import build_optimizer, solver_from_cfg
    from articulation3d_amd.modeling import build_model
    from articulation3d_amd.structures import Boxes, Instances
    from articulation3d_amd.training import DetectorTrainer
    from oracle import train_oracle as TO

    cfg = get_cfg()
    get_planercnn_cfg_defaults(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "step1_bbox.yaml"))
    cfg.MODEL.DEVICE = "cuda"
    sd = {k: v for k, v in oracle_params.items() if not k.startswith(("roi_heads.mask", "roi_heads.plane", "roi_heads.axis", "depth_head"))}

    def fresh():
        m = build_model(cfg)
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected and all("num_batches_tracked" in k for k in missing)
        return m

    frames = oracle.synthetic_frames(2, seed=77)
    tg = TO.synthetic_targets(2, seed=77)
    data = []
    for f, (b, c) in zip(frames, tg):
        inst = Instances((480, 640))
        inst.gt_boxes, inst.gt_classes = Boxes(b.cuda()), c.cuda()
        data.append({"image": torch.as_tensor(f.transpose(2, 0, 1).copy()), "instances": inst})

    model = fresh().train()
    optimizer = build_optimizer(cfg, model)
    assert optimizer.param_groups[0]["lr"] == pytest.approx(1e-3 * 1e-3)  # warm-up factor at iteration 0
    loss_dict = model(data)
    assert set(loss_dict) == {"loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"}
    losses = sum(loss_dict.values())
    optimizer.zero_grad()
    losses.backward()
    optimizer.step()
    with pytest.raises(RuntimeError):  # a re-weighted sum is refused loudly (the hand-written backward is for the plain sum)
        (2.0 * sum(model(data).values())).backward()

    ref_model = fresh().eval()
    ref = DetectorTrainer(ref_model, solver_from_cfg(cfg))
    ref_losses, _ = ref.step(torch.from_numpy(frames).cuda(), [t[0] for t in tg], [t[1] for t in tg])
    for k in loss_dict:
        assert torch.equal(loss_dict[k], ref_losses[k]), k  # same seed, same kernels: identical
    a, b = model.trainer().export_state_dict(), ref.export_state_dict()
    # (the ROIAlign backward scatters with float atomics, so two runs agree to summation-order noise, not bit for bit)
    worst = max(float((a[k] - b[k]).abs().max() / (b[k].abs().max() + 1e-12)) for k in a)
    assert worst < 1e-5, worst
    assert any(not torch.equal(a[k], sd[k].to(a[k].device)) for k in a)  # and the step did move the parameters
    before = {k: v.clone() for k, v in ref_model.state_dict().items()}
    model.eval()  # writes the trained parameters back into the modules the inference path packs from
    after = model.state_dict()
    changed = [k for k in a if not torch.equal(after[k], before[k])]
    assert len(changed) > 50 and all(torch.equal(after[k], a[k]) for k in a)
    out = model.inference_batched(torch.from_numpy(frames).cuda())
    assert out.depth is None and int(out.proposals[4].min()) > 0


def test_bf16_gradient_payload_casts_equal_torchs():
    """a3d_f32_to_bf16_scaled / a3d_bf16_to_f32 (the bf16 payload of the gradient all-reduce, parallel.allreduce_gradients) round and
    widen exactly like torch's casts (round to nearest even), incl. the 1 / world pre-division of DDP's bf16_compress_hook."""
    from articulation3d_amd import _lib

    torch.manual_seed(3)
    g = (torch.randn(1 << 20, device="cuda") * torch.logspace(-6, 3, 1 << 20, device="cuda")).contiguous()
    g16 = torch.empty(g.numel(), device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib().a3d_f32_to_bf16_scaled(g.data_ptr(), g16.data_ptr(), g.numel(), 0.125, st), "a3d_f32_to_bf16_scaled")
    assert torch.equal(g16, (g * 0.125).to(torch.bfloat16))
    back = torch.empty_like(g)
    _lib.check(_lib.lib().a3d_bf16_to_f32(g16.data_ptr(), back.data_ptr(), g.numel(), st), "a3d_bf16_to_f32")
    assert torch.equal(back, g16.to(torch.float32))


def test_bf16_storage_kernels_equal_the_fp32_storage_launch_on_the_same_values(ops, T):
    """a3d_conv_desc.io_bf16 / a3d_wgrad_desc.io_bf16: tensors STORED as bf16.  The bf16 arithmetic rounds its operands to bf16 anyway,
    so a launch on bf16-stored x / res / gate gives exactly what the fp32-storage launch gives on the same (already rounded) values,
    and a bf16-stored output is that result rounded to nearest even."""
    torch.manual_seed(8)
    for (B, H, W, Cin, Cout, k, st) in ((3, 30, 40, 256, 128, 1, 1), (2, 31, 39, 128, 128, 3, 1), (2, 30, 40, 256, 512, 1, 2)):
        x16 = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
        pk = ops.pack_conv(torch.randn(Cout, Cin, k, k) / (k * Cin ** 0.5), torch.randn(Cout) * 0.1, None, st, k // 2, ops.ACT_RELU)
        Ho, Wo = (H + 2 * (k // 2) - k) // st + 1, (W + 2 * (k // 2) - k) // st + 1
        res16 = torch.randn(B, Ho, Wo, Cout, device="cuda").to(torch.bfloat16)
        gate16 = torch.randn(B, Ho, Wo, Cout, device="cuda").to(torch.bfloat16)
        ref = ops.conv2d(x16.float(), pk, res=res16.float(), precision=1)
        assert ops.last_conv_variant().startswith("conv_bf16_kernel")
        got = ops.conv2d(x16, pk, res=res16, precision=1)                       # bf16 in, fp32 out
        assert got.dtype == torch.float32 and torch.equal(got, ref)
        got16 = ops.conv2d(x16, pk, res=res16, precision=1, out_dtype=torch.bfloat16)
        assert got16.dtype == torch.bfloat16 and torch.equal(got16, ref.to(torch.bfloat16))
        refg = ops.conv2d(x16.float(), pk, res=res16.float(), gate=gate16.float(), precision=1, act=ops.ACT_NONE)
        gotg = ops.conv2d(x16, pk, res=res16.float(), gate=gate16, precision=1, act=ops.ACT_NONE, out_dtype=torch.bfloat16)  # mixed storage
        assert torch.equal(gotg, refg.to(torch.bfloat16))
        # weight gradient with bf16-stored x / dy (and mixed)
        dy16 = torch.randn(B, Ho, Wo, Cout, device="cuda").to(torch.bfloat16)
        dw_ref = torch.empty((Cout, k * k * Cin), device="cuda")
        T.conv_wgrad(x16.float(), dy16.float(), dw_ref, KH=k, KW=k, stride=st, pad=k // 2, precision=1)
        for xa, dya in ((x16, dy16), (x16, dy16.float()), (x16.float(), dy16)):
            dw = torch.empty_like(dw_ref)
            T.conv_wgrad(xa, dya, dw, KH=k, KW=k, stride=st, pad=k // 2, precision=1)
            assert torch.equal(dw, dw_ref)
    with pytest.raises(RuntimeError):  # bf16 storage belongs to the bf16 arithmetic
        ops.conv2d(x16, pk, precision=3)
    # bias gradients (column sums) of a bf16-stored gradient = those of the widened values, bit for bit
    for M, Cc in ((5000, 256), (37, 1024), (70001, 32)):
        dy16 = torch.randn(M, Cc, device="cuda").to(torch.bfloat16)
        a, b2 = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
        T.colsum(dy16, a)
        T.colsum(dy16.float(), b2)
        assert torch.equal(a, b2)


def test_bf16_step_stores_its_resnet_activations_and_gradients_as_bf16(hip_model, oracle):
    """BASELINE configs[4] ("bf16"): in the bf16 step the ResNet stages' activations and gradients, the FPN lateral sums, the RPN hidden
    maps, the box head's hidden rows and the gradients flowing back through them live in HBM as bf16 (what torch.autocast keeps), the
    gradient all-reduce payload is bf16; master weights, the pyramid p2..p6 and its gradient, logits and losses stay fp32.
    Against the same step with fp32 storage, on the same sampled sets: RPN losses within 1 %, objectness gradients within 5 %."""
    from articulation3d_amd.training import DetectorTrainer
    from oracle import train_oracle as TO

    frames = torch.from_numpy(oracle.synthetic_frames(2)).cuda()
    tg = TO.synthetic_targets(2)
    gb, gc = [t[0] for t in tg], [t[1] for t in tg]
    tf = DetectorTrainer(hip_model, seed=5, precision="bf16", storage="fp32")
    lf, aux = tf.forward_backward(frames, gb, gc)
    rc, ri = aux["roi_count"].cpu(), aux["roi_index"].cpu()
    samples = dict(anchor_labels=aux["anchor_labels"].cpu(), roi_idx=[ri[i, : int(rc[i])].long() for i in range(2)])
    gf = {k: v.cpu() for k, v in tf.export_grads().items()}
    tb = DetectorTrainer(hip_model, seed=5, precision="bf16")
    assert tb.storage == "bf16" and tb.grad_payload == "bf16" and tf.storage == "fp32"
    lb, auxb = tb.forward_backward(frames, gb, gc, samples=samples)
    gbf = {k: v.cpu() for k, v in tb.export_grads().items()}
    assert all(t.dtype == torch.bfloat16 for t in auxb["relu_outputs"])            # res3-res5: 13 blocks x (a, b, out); RPN hidden maps; fc1 / fc2 rows
    assert all(t.dtype == torch.float32 for t in aux["relu_outputs"])
    assert auxb["feats"]["p2"].dtype == torch.float32 and tb.params.dtype == torch.float32 and tb.grads.dtype == torch.float32
    for k in ("loss_rpn_cls", "loss_rpn_loc"):
        assert abs(lb[k].item() - lf[k].item()) < 0.01 * abs(lf[k].item()) + 1e-3, (k, lb[k].item(), lf[k].item())
    rpn = [k for k in gf if k.startswith("proposal_generator.rpn_head.")]
    errs = {k: l2rel(gbf[k], gf[k]) for k in rpn}
    print("bf16 storage vs fp32 storage (bf16 arithmetic), RPN-head gradients rel L2:", {k.split("rpn_head.")[1]: round(v, 4) for k, v in errs.items()})
    assert max(v for k, v in errs.items() if "objectness" in k) < 0.05
    assert all(bool(torch.isfinite(v).all()) for v in gbf.values())
    for k in gf:
        assert 0.5 < float(gbf[k].norm() / (gf[k].norm() + 1e-30)) < 2.0, k


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_batched_launches_give_the_per_layer_launches_bits(hip_model, oracle, precision):
    """Round 4: ONE launch for the data-gradient filters of all layers (a3d_weight_transpose_batch) and ONE for the slice reductions of
    the step's weight gradients (a3d_wgrad_reduce_batch) instead of one per layer.  Same sums in the same order: on the same batch and
    sampled sets EVERY gradient agrees BIT FOR BIT (and two runs of one form do: no float atomics are left in the step), and the
    transposed filters are identical."""
    from articulation3d_amd import training
    from articulation3d_amd.training import DetectorTrainer
    from oracle import train_oracle as TO

    frames = torch.from_numpy(oracle.synthetic_frames(2)).cuda()
    tg = TO.synthetic_targets(2)
    gb, gc = [t[0] for t in tg], [t[1] for t in tg]
    saved = training.BATCHED_LAUNCHES
    try:
        training.BATCHED_LAUNCHES = False
        t0 = DetectorTrainer(hip_model, seed=5, precision=precision)
        l0, aux = t0.forward_backward(frames, gb, gc)
        rc, ri = aux["roi_count"].cpu(), aux["roi_index"].cpu()
        samples = dict(anchor_labels=aux["anchor_labels"].cpu(), roi_idx=[ri[i, : int(rc[i])].long() for i in range(2)])
        g0 = {k: v.clone() for k, v in t0.export_grads().items()}
        wt0 = t0._wt.clone()
        t0.forward_backward(frames, gb, gc, samples=samples)  # the per-layer form once more: its own run-to-run spread (float atomics)
        spread = {k: l2rel(v, g0[k]) for k, v in t0.export_grads().items()}
        training.BATCHED_LAUNCHES = True
        t1 = DetectorTrainer(hip_model, seed=5, precision=precision)
        for _ in range(2):  # (twice: the second step reuses the cached device tables and the persistent workspaces)
            l1, _ = t1.forward_backward(frames, gb, gc, samples=samples)
        g1 = t1.export_grads()
        assert t1._defer is not None and t1._tbatch is not None and not t1._defer.items
    finally:
        training.BATCHED_LAUNCHES = saved
    assert torch.equal(t1._wt, wt0)
    for k in l0:
        assert float(l0[k]) == float(l1[k]), k
    # (round 4, later: the ROIAlign backward sums in a fixed order too -- tile-gather form, no float atomics -- so the WHOLE step is
    # bit-reproducible: the per-layer form's two runs agree exactly, and so do the two forms, on every gradient)
    assert max(spread.values()) == 0.0, {k: v for k, v in spread.items() if v > 0}
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k


def test_roi_align_backward_gather_at_the_training_step_size(T):
    """BASELINE configs[4]'s shapes (16 images x 512 sampled ROIs, 7x7 bins, 256 channels, the four pyramid levels of a 480x640 frame), boxes
    clustered around a few objects as the sampler leaves them: the tile-gather form against the float-atomics form (fp32 rounding of the
    sums) and against itself (bit for bit) -- the property that makes the whole training step reproducible."""
    g = torch.Generator().manual_seed(5)
    B, R, C = 16, 512, 256
    ctr = torch.rand(B, 6, 2, generator=g) * torch.tensor([560.0, 400.0]) + 40
    pick = torch.randint(0, 6, (B, R), generator=g)
    c = torch.gather(ctr, 1, pick[:, :, None].expand(B, R, 2)) + torch.randn(B, R, 2, generator=g) * 25
    wh = torch.rand(B, R, 2, generator=g) * 380 + 20
    boxes = torch.cat([c - wh / 2, c + wh / 2], 2).clamp(min=0)
    boxes[..., 2].clamp_(max=640)
    boxes[..., 3].clamp_(max=480)
    boxes = boxes.cuda().contiguous()
    dout = torch.randn(B * R, 7, 7, C, device="cuda", generator=torch.Generator(device="cuda").manual_seed(6))
    mk = lambda: [torch.zeros(B, 480 // s, 640 // s, C, device="cuda") for s in (4, 8, 16, 32)]
    a, a2, sc = mk(), mk(), mk()
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    T.roi_align_fpn_backward(a, scales, boxes, dout, P=7, sampling_ratio=0, aligned=True)
    T.roi_align_fpn_backward(a2, scales, boxes, dout, P=7, sampling_ratio=0, aligned=True)
    T.roi_align_fpn_backward(sc, scales, boxes, dout, P=7, sampling_ratio=0, aligned=True, scatter=True)
    assert all(torch.equal(p, q) for p, q in zip(a, a2))
    num = sum(float((p.double() - q.double()).pow(2).sum()) for p, q in zip(a, sc)) ** 0.5
    den = sum(float(q.double().pow(2).sum()) for q in sc) ** 0.5
    assert num / den < 1e-6, num / den
    assert all(bool(torch.isfinite(p).all()) for p in a) and den > 0


@pytest.mark.parametrize("case", [(4, 60, 80, 256, 256, 3, 1, "bf16", "f32", False),   # FPN output conv: bf16-stored lateral sums in, fp32 pyramid out
                                  (4, 60, 80, 256, 256, 3, 1, "f32", "bf16", False),   # RPN conv: fp32 pyramid in (converted on the fragment), bf16 out
                                  (3, 61, 79, 128, 128, 3, 1, "bf16", "bf16", False),  # ragged pixel tile, one 128-channel tile
                                  (6, 30, 40, 1024, 256, 1, 1, "bf16", "bf16", True),  # deep 1x1 + residual
                                  (2, 60, 80, 256, 384, 3, 2, "bf16", "bf16", False),  # strided, ragged channel tile
                                  (1, 1, 700, 12544, 1024, 1, 1, "bf16", "bf16", False)],  # fc1 rows
                         ids=lambda c: "x".join(str(v) for v in c))
def test_bf16_dma_kernel_equals_the_register_staged_one_bit_for_bit(ops, case):
    """Round 5: conv_bf16w_kernel (csrc/conv_bf16w.hip) -- the training step's bf16 launches with BOTH operands by LDS-DMA (the filter from
    the trainer's per-step bf16 copy, a3d_conv_desc.w_bf16), 256-pixel tiles, ping-pong halves -- against conv_bf16_kernel, which rounds
    the fp32 filter on the fly: the same rounded operands, the same chunk and step order -> equal bits, for bf16- and fp32-stored
    activations and outputs, with residual and the ReLU-backward gate, 3x3 taps in the zero padding, strided and ragged tiles."""
    B, H, W, Cin, Cout, k, st, xs, os_, has_res = case
    dt = {"bf16": torch.bfloat16, "f32": torch.float32}
    torch.manual_seed(31)
    x = torch.randn(B, H, W, Cin, device="cuda").to(dt[xs])
    pk = ops.pack_conv(torch.randn(Cout, Cin, k, k) / (k * Cin ** 0.5), torch.randn(Cout) * 0.1, None, st, k // 2, ops.ACT_RELU)
    Ho, Wo = (H + 2 * (k // 2) - k) // st + 1, (W + 2 * (k // 2) - k) // st + 1
    res = torch.randn(B, Ho, Wo, Cout, device="cuda").to(dt[os_]) if has_res else None
    gate = torch.randn(B, Ho, Wo, Cout, device="cuda").to(torch.bfloat16)
    old = ops.conv2d(x, pk, res=res, precision=1, out_dtype=dt[os_])
    assert ops.last_conv_variant().startswith("conv_bf16_kernel"), ops.last_conv_variant()
    pk.w_b16 = pk.w.to(torch.bfloat16)  # (the trainer's per-step copy: a3d_f32_to_bf16_scaled rounds like torch, test_bf16_gradient_payload_casts_equal_torchs)
    for tune in (30, 31):
        new = ops.conv2d(x, pk, res=res, precision=1, out_dtype=dt[os_], tune=tune)
        assert ops.last_conv_variant() == f"conv_bf16w_kernel<{2 if tune == 30 else 4}>", ops.last_conv_variant()
        assert new.dtype == old.dtype and torch.equal(new, old), (tune, float((new.float() - old.float()).abs().max()))
    if os_ == "bf16":  # the data-gradient form: gate + no activation
        pk.w_b16 = None
        og = ops.conv2d(x, pk, gate=gate, precision=1, act=ops.ACT_NONE, out_dtype=torch.bfloat16)
        pk.w_b16 = pk.w.to(torch.bfloat16)
        ng = ops.conv2d(x, pk, gate=gate, precision=1, act=ops.ACT_NONE, out_dtype=torch.bfloat16, tune=30)
        assert ops.last_conv_variant() == "conv_bf16w_kernel<2>" and torch.equal(ng, og)


@pytest.mark.parametrize("case", [(4, 60, 80, 128, 512, "bf16", "bf16", True, True),    # data gradient of a bottleneck's conv1: residual + gate, all stored bf16
                                  (4, 60, 80, 128, 512, "bf16", "bf16", True, False),   # its forward twin conv3: residual, ReLU
                                  (3, 31, 39, 256, 1024, "f32", "bf16", False, True),   # fp32-stored input (rounded on the way into the fragments), ragged last pixel tile
                                  (2, 120, 160, 32, 256, "f32", "bf16", False, True),   # the RPN predictors' data gradient: one ring step per N step
                                  (5, 60, 80, 512, 128, "bf16", "bf16", False, True),   # Cin 512: 128 fragment registers, 64-channel N steps
                                  (3, 60, 80, 512, 256, "bf16", "f32", False, False),   # fp32 output, nothing beside it
                                  (1, 37, 53, 256, 512, "bf16", "bf16", True, True)],   # below the size rule (forced): partial round split along N
                         ids=lambda c: "x".join(str(v) for v in c))
def test_bf16_pointwise_kernel_with_stationary_activations_equals_the_tiled_one_bit_for_bit(ops, case):
    """Round 5: conv_bf16xs_kernel (csrc/conv_bf16xs.hip) -- the step's HBM-bound 1x1 launches with a wave's 32 pixels held as MFMA fragments for
    the whole launch, the bf16 filter copy streamed by LDS-DMA, residual and gate rows requested an N step ahead of their use -- against
    conv_bf16_kernel: the same rounded operands and the same 16-deep products in the same order -> equal bits."""
    B, H, W, Cin, Cout, xs, os_, has_res, has_gate = case
    dt = {"bf16": torch.bfloat16, "f32": torch.float32}
    torch.manual_seed(37)
    x = torch.randn(B, H, W, Cin, device="cuda").to(dt[xs])
    pk = ops.pack_conv(torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5, torch.randn(Cout) * 0.1, None, 1, 0, ops.ACT_NONE if has_gate else ops.ACT_RELU)
    res = torch.randn(B, H, W, Cout, device="cuda").to(torch.bfloat16) if has_res else None
    gate = torch.randn(B, H, W, Cout, device="cuda").to(torch.bfloat16) if has_gate else None
    kw = dict(res=res, gate=gate, precision=1, out_dtype=dt[os_])
    old = ops.conv2d(x, pk, **kw)
    assert ops.last_conv_variant().startswith("conv_bf16_kernel"), ops.last_conv_variant()
    pk.w_b16 = pk.w.to(torch.bfloat16)
    new = ops.conv2d(x, pk, tune=33, **kw)
    assert ops.last_conv_variant() == f"conv_bf16xs_kernel<{Cin}>", ops.last_conv_variant()
    assert new.dtype == old.dtype and torch.equal(new, old), float((new.float() - old.float()).abs().max())
    assert bool(torch.isfinite(new.float()).all()) and float(new.float().abs().max()) > 0
    kept = ops.conv2d(x, pk, tune=34, **kw)  # tune 34: never this kernel
    assert not ops.last_conv_variant().startswith("conv_bf16xs") and torch.equal(kept, old)


@pytest.mark.parametrize("precision", ["bf16", "bf16x3"])
def test_side_streams_keep_every_bit_of_the_serial_step(hip_model, oracle, precision):
    """Round 5: the weight gradients run on a side stream beside the data-gradient chain, and the RPN head's backward is enqueued on a second
    stream as soon as its loss exists (under the proposal selection / sampling window); the pooler's backward then adds ONTO the RPN term
    of every pyramid level instead of the other way round (a + b either way, p6's term last in both).  Against the one-stream step on the
    same batch and sampled sets: every loss and every gradient bit for bit, step after step."""
    from articulation3d_amd.training import DetectorTrainer
    from oracle import train_oracle as TO

    frames = torch.from_numpy(oracle.synthetic_frames(2)).cuda()
    tg = TO.synthetic_targets(2)
    gb, gc = [t[0] for t in tg], [t[1] for t in tg]
    t0 = DetectorTrainer(hip_model, seed=7, precision=precision)
    assert t0._wg_stream is not None and t0._rpn_stream is not None  # (the defaults under test)
    t0._wg_stream = t0._rpn_stream = None  # one stream
    l0, aux = t0.forward_backward(frames, gb, gc)
    rc, ri = aux["roi_count"].cpu(), aux["roi_index"].cpu()
    samples = dict(anchor_labels=aux["anchor_labels"].cpu(), roi_idx=[ri[i, : int(rc[i])].long() for i in range(2)])
    l0, _ = t0.forward_backward(frames, gb, gc, samples=samples)
    g0 = {k: v.clone() for k, v in t0.export_grads().items()}
    t1 = DetectorTrainer(hip_model, seed=7, precision=precision)
    for _ in range(3):  # (later steps reuse the events and the allocator's blocks of both side streams)
        l1, _ = t1.forward_backward(frames, gb, gc, samples=samples)
        torch.cuda.synchronize()
        g1 = t1.export_grads()
        for k in l0:
            assert float(l0[k]) == float(l1[k]), k
        for k in g0:
            assert torch.equal(g0[k], g1[k]), k
    assert all(bool(torch.isfinite(v).all()) for v in g0.values()) and sum(float(v.abs().sum()) for v in g0.values()) > 0


def test_bf16_pointwise_kernel_fuzz_against_the_tiled_one(ops):
    """Random shapes through conv_bf16xs_kernel (tune 33) against conv_bf16_kernel (tune 34): ragged pixel tiles, every Cin it takes, channel
    counts that split the last round along N in different ways, each storage / residual / gate combination -- equal bits in every case."""
    import random

    rng = random.Random(5)
    bf, f32 = torch.bfloat16, torch.float32
    ran = 0
    for it in range(120):
        Cin = rng.choice([32, 128, 256, 512])
        bn = 64 if Cin == 512 else 128
        Cout = bn * rng.choice([1, 2, 3, 4, 8])
        B, H, W = rng.choice([1, 2, 3]), rng.randint(5, 70), rng.randint(5, 90)
        xdt = rng.choice([bf, f32]) if Cin != 512 else bf
        odt = rng.choice([bf, f32])
        has_res, has_gate = rng.random() < 0.5, rng.random() < 0.5
        torch.manual_seed(1000 + it)
        x = torch.randn(B, H, W, Cin, device="cuda").to(xdt)
        pk = ops.pack_conv(torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5, torch.randn(Cout) * 0.1, None, 1, 0, ops.ACT_NONE if has_gate else ops.ACT_RELU)
        pk.w_b16 = pk.w.to(bf)
        res = torch.randn(B, H, W, Cout, device="cuda").to(bf) if has_res else None
        gate = torch.randn(B, H, W, Cout, device="cuda").to(bf) if has_gate else None
        kw = dict(res=res, gate=gate, precision=1, out_dtype=odt)
        old = ops.conv2d(x, pk, tune=34, **kw)
        assert ops.last_conv_variant().startswith("conv_bf16_kernel"), ops.last_conv_variant()
        new = ops.conv2d(x, pk, tune=33, **kw)
        assert ops.last_conv_variant() == f"conv_bf16xs_kernel<{Cin}>", (ops.last_conv_variant(), (B, H, W, Cin, Cout))
        assert torch.equal(new, old), ((B, H, W, Cin, Cout, xdt, odt, has_res, has_gate), float((new.float() - old.float()).abs().max()))
        ran += 1
    assert ran == 120


# ------------------------------------------------------------------------------------------ gradient exchange under the backward pass
def _exchange_check(args, timeout=1500):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "tools/train_exchange_check.py", *args], cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_gradient_segments_leave_during_the_backward_pass_bit_identical_one_rccl_rank():
    """VERDICT r5 item 2.  `DetectorTrainer.step` hands the flat gradient buffer to the collective in four segments (box head | res5 +
    FPN + RPN head | res4 | res3), each the moment its last weight gradient has been enqueued -- on the real RCCL process group with one
    rank (the 1-GPU box).  Against the same segments announced only BEHIND the whole backward pass: parameters, momenta and losses after
    three steps bit-identical, for both payloads.  (The bf16 payload rewrites every segment with what the communication stream read: a
    segment announced before its last weight gradient landed would differ.)"""
    d = _exchange_check(["--mode", "world1", "--backend", "nccl", "--steps", "3"])
    for payload in ("fp32", "bf16"):
        r = d["result"][payload]
        assert r["params_equal"] and r["momentum_equal"] and r["losses_equal"] and r["finite"], (payload, r)
        assert r["segments_per_step"] == [4, 4] and r["forms"] == ["force", "force-late"], r
    assert d["result"]["bf16"]["payload_bytes_per_step"] * 2 == d["result"]["fp32"]["payload_bytes_per_step"] > 160e6


def test_gradient_segments_equal_the_monolithic_allreduce_two_gloo_ranks():
    """Two ranks (gloo: both on this box's one GPU, different batches): the overlapped segmented exchange against ONE all-reduce behind the
    backward pass (`A3D_TRAIN_GRAD_OVERLAP=0`, parallel.allreduce_gradients) -- parameters after three steps bit-identical, fp32 and
    bf16 payloads (tools/train_net.py:96,110-117: DDP's bucketed, overlapped all-reduce)."""
    d = _exchange_check(["--mode", "world2", "--backend", "gloo", "--steps", "3"])
    assert d["world"] == 2
    for payload in ("fp32", "bf16"):
        r = d["result"][payload]
        assert r["params_equal"] and r["momentum_equal"] and r["losses_equal"] and r["finite"], (payload, r)
        assert r["segments_per_step"] == [4, 0], r
