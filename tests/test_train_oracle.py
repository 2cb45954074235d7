"""CPU suite: the training-step oracle (oracle/train_oracle.py) against independent statements of the same
detectron2 rules, and the host-side pieces of the product trainer (sampling, learning-rate schedule)."""
import math

import pytest
import torch
import torch.nn.functional as F


@pytest.fixture(scope="module")
def TO():
    from oracle import train_oracle

    return train_oracle


def test_pairwise_iou_and_matcher_rules(TO):
    gt = torch.tensor([[0.0, 0.0, 10.0, 10.0], [20.0, 20.0, 40.0, 40.0]])
    boxes = torch.tensor([[0.0, 0.0, 10.0, 10.0],      # IoU 1 with gt0
                          [0.0, 0.0, 10.0, 5.0],       # 0.5 with gt0
                          [100.0, 100.0, 110.0, 110.0],  # no overlap
                          [22.0, 22.0, 38.0, 38.0],    # inside gt1: 256/400 = 0.64 (best for gt1, below 0.7)
                          [0.0, 0.0, 10.0, 2.0]])      # 0.2 with gt0
    q = TO.pairwise_iou(gt, boxes)
    assert torch.allclose(q[0], torch.tensor([1.0, 0.5, 0.0, 0.0, 0.2]))
    assert abs(q[1, 3].item() - 0.64) < 1e-6
    idx, lab = TO.matcher(q, (0.3, 0.7), (0, -1, 1), True)
    assert idx.tolist() == [0, 0, 0, 1, 0]
    assert lab.tolist() == [1, -1, 0, 1, 0]  # box 3 is a low-quality match (best anchor of gt1)
    idx, lab = TO.matcher(q, (0.3, 0.7), (0, -1, 1), False)
    assert lab.tolist() == [1, -1, 0, -1, 0]
    idx, lab = TO.matcher(TO.pairwise_iou(gt[:0], boxes), (0.5,), (0, 1), False)  # image without ground truth
    assert idx.tolist() == [0] * 5 and lab.tolist() == [0] * 5


def test_get_deltas_inverts_apply_deltas(TO, oracle):
    torch.manual_seed(0)
    src = torch.rand(50, 4) * 100
    src[:, 2:] += src[:, :2] + 5
    tgt = torch.rand(50, 4) * 100
    tgt[:, 2:] += tgt[:, :2] + 5
    for w in ((1.0, 1.0, 1.0, 1.0), (10.0, 10.0, 5.0, 5.0)):
        d = TO.get_deltas(src, tgt, w)
        back = oracle.apply_deltas(d, src, w, math.log(1000.0 / 16))
        assert torch.allclose(back, tgt, atol=1e-3)


def test_subsample_labels_counts_and_product_sampler_agrees(TO):
    from articulation3d_amd.training import subsample_labels as product_sampler

    labels = torch.full((5000,), -1, dtype=torch.int8)
    labels[:40] = 1
    labels[100:3000] = 0
    pos, neg = TO.subsample_labels(labels, 256, 0.5, 0, torch.Generator().manual_seed(3))
    assert len(pos) == 40 and len(neg) == 216 and (labels[pos] == 1).all() and (labels[neg] == 0).all()
    p2, n2 = product_sampler(labels, 256, 0.5, 0, torch.Generator().manual_seed(3))  # same draws from the same seed
    assert torch.equal(pos, p2) and torch.equal(neg, n2)
    labels[:400] = 1
    pos, neg = TO.subsample_labels(labels, 256, 0.5, 0, torch.Generator().manual_seed(3))
    assert len(pos) == 128 and len(neg) == 128


def test_losses_against_plain_torch(TO, oracle):
    cfg, tc = oracle.OracleCfg(), TO.TrainCfg()
    torch.manual_seed(1)
    M, K = 64, cfg.num_classes
    scores, deltas = torch.randn(M, K + 1), torch.randn(M, 4 * K)
    cls = torch.randint(0, K + 1, (M,))
    boxes = torch.rand(M, 4) * 50
    boxes[:, 2:] += boxes[:, :2] + 4
    gt = boxes + 1.5
    out = TO.box_losses(scores, deltas, boxes, cls, gt, cfg)
    want_cls = -(F.log_softmax(scores, 1)[torch.arange(M), cls]).mean()
    assert abs(out["loss_cls"].item() - want_cls.item()) < 1e-6
    tot = 0.0
    tgt = TO.get_deltas(boxes, gt, cfg.box_weights)
    for r in range(M):
        if cls[r] < K:
            tot += (deltas[r, 4 * cls[r]: 4 * cls[r] + 4] - tgt[r]).abs().sum().item()
    assert abs(out["loss_box_reg"].item() - tot / M) < 1e-5


def test_roi_align_backward_is_the_adjoint_of_forward(TO, oracle):
    """<roi_align(f), g> == <f, roi_align_backward(g)> for the C forward / backward pair."""
    torch.manual_seed(2)
    feats = {n: torch.randn(1, 8, 480 // s, 640 // s, requires_grad=True) for n, s in (("p2", 4), ("p3", 8), ("p4", 16), ("p5", 32))}
    boxes = torch.tensor([[10.0, 20.0, 200.0, 300.0], [300.0, 100.0, 340.0, 130.0], [-5.0, -5.0, 650.0, 490.0], [50.0, 60.0, 50.4, 60.3]])
    pooled = TO.roi_pool_fpn_diff(feats, [boxes], 7, 0, True)
    g = torch.randn_like(pooled)
    pooled.backward(g)
    lhs = (pooled.detach().double() * g.double()).sum().item()
    rhs = sum((f.detach().double() * f.grad.double()).sum().item() for f in feats.values() if f.grad is not None)
    assert abs(lhs - rhs) < 1e-4 * abs(lhs)


def test_lr_schedule_and_sgd(TO):
    from articulation3d_amd.training import SolverCfg, lr_at

    tc, sc = TO.TrainCfg(), SolverCfg()
    for it in (0, 1, 500, 999, 1000, 209999, 210000, 250000):
        assert lr_at(it, sc) == TO.lr_at(it, tc)
    assert abs(TO.lr_at(0, tc) - 1e-6) < 1e-12 and TO.lr_at(1000, tc) == 1e-3 and abs(TO.lr_at(250000, tc) - 1e-5) < 1e-12
    # torch.optim.SGD is the rule being restated
    w = torch.nn.Parameter(torch.randn(20))
    opt = torch.optim.SGD([w], lr=0.01, momentum=tc.momentum, weight_decay=tc.weight_decay)
    P, bufs = {"w": w.detach().clone()}, {}
    for _ in range(3):
        g = torch.randn(20)
        w.grad = g.clone()
        opt.step()
        TO.sgd_step(P, {"w": g}, bufs, 0.01, tc)
        assert torch.allclose(P["w"], w.detach(), atol=1e-7)


def test_trainable_parameter_set(TO, oracle_params):
    names = TO.trainable_names(oracle_params)
    assert sum(oracle_params[k].numel() for k in names) == 41077786  # ~41.1 M trainable under FREEZE_AT 2
    assert not any(".norm." in k or ".stem." in k or ".res2." in k for k in names)
    assert not any(k.startswith(("roi_heads.mask_head", "roi_heads.plane_head", "roi_heads.axis_head", "depth_head")) for k in names)
