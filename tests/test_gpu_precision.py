"""GPU suite (-m gpu): the DEFAULT ARITHMETIC under adversarial inputs.

fp16x2 (a3d_conv_desc.precision == 3) is a block-floating-point format: every operand is represented by two fp16 terms
(22 significand bits) under ONE power-of-two exponent per image (per ROI behind the poolers, per layer for filters).  What that
guarantees, elementwise, for an output y = sum_k x_k w_k of image b (A = max |x[b]|, Wmax = max |w|):

    | y - y64 |  <=  c * ( 2^-22 * sum_k |x_k| |w_k|                                   -- the fp32-grade term
                          + 2^-40 * ( A * sum_k |w_k| + Wmax * sum_k |x_k| ) )         -- the block exponent's absolute floor

The first term is what fp32 itself delivers (its own constant is ~2^-24 sqrt(K)); the second is negligible unless an output's
whole receptive field lies more than 2^18 below its image's maximum -- THE WINDOW.  These tests
  * assert the two-term law on every output for outliers up to 2^30 x the rest of the image,
  * assert the fp32-style ONE-term law wherever the receptive field is inside the window (outlier ratios up to 2^17), and locate
    the ratio at which it stops holding (>= 2^18: the documented window, not earlier),
  * pin what the code does about the window: ROI rows are scaled by their OWN maximum (recorded by the pooler), and ROIs fainter
    than 2^-16 of their pyramid level are counted by a device-side monitor (ops.roi_window_count) instead of passing silently,
  * pin the non-finite semantics: a NaN poisons exactly the outputs fp32 poisons (an Inf: its whole receptive field, a superset of
    fp32's), the finite values of its image keep their scale, other images do not change by a bit,
  * pin the >= 4 GiB batches (block-wise launches) and the per-image maxima of arbitrary tensors.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from articulation3d_amd import ops as o

    return o


def _ref64(x, w, bias, stride, pad):
    """float64 convolution + the two sums of the error law.  x NHWC fp32 (cuda), w [Cout,Cin,k,k] fp32 (cpu)."""
    xd = x.permute(0, 3, 1, 2).double()
    wd = w.double().cuda()
    y = F.conv2d(xd, wd, None if bias is None else bias.double().cuda(), stride=stride, padding=pad)
    s = F.conv2d(xd.abs(), wd.abs(), None, stride=stride, padding=pad)                      # sum |x||w|
    w1 = wd.abs().sum((1, 2, 3))                                                             # sum |w| per output channel
    x1 = F.conv2d(xd.abs().sum(1, keepdim=True), torch.ones(1, 1, w.shape[2], w.shape[3], dtype=torch.float64, device="cuda"), None,
                  stride=stride, padding=pad)                                               # sum |x| over the receptive field
    return y.permute(0, 2, 3, 1), s.permute(0, 2, 3, 1), w1, x1.permute(0, 2, 3, 1)


LAW_CASES = [  # name, (B, H, W, Cin, Cout, k, stride), conv2d kwargs, expected kernel prefix, winograd?
    ("direct-1x1", (2, 24, 40, 256, 256, 1, 1), {}, "conv_h2sg_kernel", False),   # (480 tiles of 32 x 32: the small-grid form, csrc/conv_sg_h2.hip)
    ("direct-1x1-tiled", (2, 24, 40, 256, 256, 1, 1), dict(tune=11), "conv_h2_kernel", False),
    ("direct-3x3", (2, 24, 40, 128, 128, 3, 1), {}, "conv_h2_kernel", False),
    ("direct-3x3-s2", (2, 25, 39, 128, 128, 3, 2), {}, "conv_h2_kernel", False),
    ("wide-fc", (600, 1, 1, 4096, 512, 1, 1), dict(tune=9), "conv_h2w_kernel", False),
    ("winograd", (2, 24, 40, 256, 256, 3, 1), {}, "wino_gemm_h2w_kernel", True),
]
RATIOS_LOG2 = (10, 14, 17, 18, 20, 24, 30)
def c_law(K):
    """Constant of the law: 8, plus fp32 ACCUMULATION's own sqrt(K) * 2^-24 = sqrt(K) / 4 in units of 2^-22 (any fp32 evaluation of a
    K-term sum carries it; measured maxima of err / bound are printed: 0.1-0.6)."""
    return 8.0 + math.sqrt(K) / 4.0


@pytest.mark.parametrize("case", LAW_CASES, ids=lambda c: c[0])
def test_fp16x2_elementwise_error_law_and_its_window(ops, case):
    name, (B, H, W, Cin, Cout, k, st), kw, kernel, wino = case
    torch.manual_seed(31)
    w = torch.randn(Cout, Cin, k, k) / (k * Cin ** 0.5)
    pk = ops.pack_conv(w, None, None, st, k // 2, ops.ACT_NONE)
    base = torch.randn(B, H, W, Cin, device="cuda")
    wmax = float(w.abs().max())
    C_LAW = c_law(Cin * k * k)
    first_fail = None
    worst2 = 0.0
    for r in RATIOS_LOG2:
        x = base.clone()
        if H > 1:
            x[0, H // 2, W // 2, :] *= 2.0 ** r  # one hot cell (all its channels) in image 0; image 1.. stay at unit scale
        else:
            x[0, 0, 0, :8] *= 2.0 ** r  # FC rows are their own "images": eight hot features in row 0
        y = ops.conv2d(x, pk, precision=3, **kw)
        assert ops.last_conv_variant().startswith(kernel), ops.last_conv_variant()
        y64, s, w1, x1 = _ref64(x, w, None, st, k // 2)
        err = (y[..., :Cout].double() - y64).abs()
        A = x.abs().flatten(1).amax(1).double()[:, None, None, None]
        if wino:
            # Winograd computes a 2x2 output tile from its whole 4x4 input patch: the natural elementwise scale of an output is the
            # patch's sum |x| |w| (fp32 Winograd has the same property), i.e. the 3x3 sums max-pooled over the tile's neighbourhood
            s = F.max_pool2d(s.permute(0, 3, 1, 2), 3, 1, 1).permute(0, 2, 3, 1)
            x1 = F.max_pool2d(x1.permute(0, 3, 1, 2), 3, 1, 1).permute(0, 2, 3, 1)
        # absolute floor of the second fp16 term: half an fp16 subnormal ulp (2^-25) over a scale of at least 2^14 / A, i.e. 2^-39 A per
        # element (Winograd: the V rows are scaled for 4 A and sum |G g G^T| <= 4 sum |g|)
        floor = (16.0 if wino else 1.0) * 2.0 ** -39 * (A * w1[None, None, None, :] + wmax * x1)
        one = C_LAW * 2.0 ** -22 * s
        two = one + C_LAW * floor
        ratio2 = float((err / (two + 1e-300)).max())
        worst2 = max(worst2, ratio2)
        assert ratio2 <= 1.0, (name, r, "two-term law", ratio2)
        ok_one = bool((err <= one + 1e-300).all())
        # images without the outlier are inside the window by construction: the fp32-style law always holds there
        assert bool((err[1:] <= one[1:] + 1e-300).all()), (name, r, "clean images")
        if r <= 17:
            assert ok_one, (name, r, "one-term law inside the window", float((err / (one + 1e-300)).max()))
        elif not ok_one and first_fail is None:
            first_fail = r
    print(f"\n{name}: two-term law holds to 2^30 (max err/bound {worst2:.3f}); fp32-style one-term law first violated at outlier ratio 2^{first_fail}")
    assert first_fail is None or first_fail >= 18


def test_roi_rows_are_scaled_by_their_own_maximum_and_faint_rois_are_counted(ops):
    """A pyramid level with one hot cell 10^6 x the rest.  (1) The pooler records every ROI's OWN maximum (exactly), so a ROI far from
    the hot cell is split at its own scale: its head convolution obeys the fp32-style one-term law.  (2) The window monitor counts
    that ROI (its level maximum is > 2^16 x its own), and does not count it when the hot cell is only 10^4 x."""
    torch.manual_seed(5)
    C = 256
    feats = [torch.randn(1, 120 // (1 << l), 160 // (1 << l), C, device="cuda") for l in range(4)]
    boxes = torch.tensor([[[40.0, 40, 130, 120], [300, 200, 390, 290], [250, 180, 300, 260]]], device="cuda")  # sqrt(area) < 112: level p2
    count = torch.tensor([3], device="cuda", dtype=torch.int32)
    w = torch.randn(C, C, 3, 3) / (3 * C ** 0.5)
    pk = ops.pack_conv(w, None, None, 1, 1, ops.ACT_NONE)
    for hot, counted in ((1e6, True), (1e4, False)):
        f = [t.clone() for t in feats]
        f[0][0, 60, 80, :] = hot  # inside box 1 (y 50..72, x 75..97 at stride 4), outside boxes 0 and 2
        for t in f:
            t._a3d_amax = t.abs().flatten(1).amax(1)
        n0 = ops.roi_window_count()
        pooled = ops.roi_align_fpn(f, [1 / 4, 1 / 8, 1 / 16, 1 / 32], boxes, count, 14, 0, False)
        torch.cuda.synchronize()
        assert torch.equal(pooled._a3d_amax[:3], pooled[:3].abs().flatten(1).amax(1))  # the ROI's own maximum, exactly
        assert float(pooled._a3d_amax[1]) > 1e3 and float(pooled._a3d_amax[0]) < 10.0
        assert (ops.roi_window_count() - n0 >= 1) == counted, (hot, ops.roi_window_count() - n0)
        y = ops.conv2d(pooled, pk, precision=3)
        assert ops.last_conv_variant().startswith("wino_gemm_h2w_kernel")
        y64, s, w1, x1 = _ref64(pooled, w, None, 1, 1)
        s = F.max_pool2d(s.permute(0, 3, 1, 2), 3, 1, 1).permute(0, 2, 3, 1)
        err = (y[..., :C].double() - y64).abs()
        assert bool((err[0] <= c_law(9 * C) * 2.0 ** -22 * s[0]).all()), float((err[0] / s[0]).max())  # the faint ROI, at ITS scale


NONFINITE_CASES = [
    ("direct-1x1", (3, 16, 20, 256, 128, 1, 1), {}, False),
    ("direct-3x3", (3, 16, 20, 128, 128, 3, 1), {}, False),
    ("direct-3x3-s2", (3, 17, 21, 128, 128, 3, 2), {}, False),
    ("winograd", (3, 16, 20, 256, 256, 3, 1), {}, True),
    ("fc", (300, 1, 1, 1024, 256, 1, 1), {}, False),
]


@pytest.mark.parametrize("precision", [3, 2, 0], ids=["fp16x2", "bf16x3", "fp32"])
@pytest.mark.parametrize("case", NONFINITE_CASES, ids=lambda c: c[0])
def test_nan_and_inf_poison_what_fp32_poisons_and_nothing_else(ops, case, precision):
    """Image 1 holds one NaN, image 2 one +Inf (detectron2 only filters non-finite candidates AFTER the network:
    pkg/modeling/meta_arch/planercnn.py:168,176 -> find_top_rpn_proposals / fast_rcnn_inference).  In every fp32-grade mode:
      * the clean image's output does not change by a bit, and neither does any output of a poisoned image outside the receptive
        field of the poisoned position (the image keeps its scale: the recorded maxima ignore non-finite values);
      * everything a plain fp32 convolution turns non-finite is non-finite, and nothing outside the receptive field is (Winograd
        layers: outside the 2x2 tiles whose 4x4 input patch holds the value -- the form's own footprint, in fp32 too);
      * NaN: EXACTLY fp32's set.  Inf: exactly fp32's set in the fp32-input mode; the split-operand modes represent Inf as
        h + l = Inf + (Inf - Inf) and therefore turn its whole receptive field into NaN, where fp32 has +Inf / -Inf and the ReLU
        maps the -Inf half back to a finite 0 -- a superset, confined to the same receptive field."""
    name, (B, H, W, Cin, Cout, k, st), kw, wino = case
    torch.manual_seed(17)
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, k, k) / (k * Cin ** 0.5)
    bias = torch.randn(Cout) * 0.1
    pk = ops.pack_conv(w, bias, None, st, k // 2, ops.ACT_RELU)
    py, px = (H // 2, W // 3) if H > 1 else (0, 0)
    x[1, py, px, 5] = 0.25
    x[2, py, px, 7] = -0.25
    clean = ops.conv2d(x, pk, precision=precision, **kw).clone()
    xp = x.clone()
    xp[1, py, px, 5] = float("nan")
    xp[2, py, px, 7] = float("inf")
    y = ops.conv2d(xp, pk, precision=precision, **kw)
    torch.cuda.synchronize()
    ref = F.relu(F.conv2d(xp.permute(0, 3, 1, 2), w.cuda(), bias.cuda(), stride=st, padding=k // 2)).permute(0, 2, 3, 1)
    assert torch.equal(y[0], clean[0])  # the clean image: bit-unchanged
    bad, bad_ref = ~torch.isfinite(y[..., :Cout]), ~torch.isfinite(ref)
    assert not bool(bad[0].any())
    uses_wino = ops.last_conv_variant().startswith("wino")  # (the fp32-input mode takes the Winograd form for every 3x3 s1 layer)
    # the receptive field of the poisoned input position = where fp32 turns the NaN image's outputs into NaN (every channel)
    touched = bad_ref[1].any(-1)
    if uses_wino:  # the form's own footprint (also in fp32): the 2x2 tiles whose 4x4 input patch (rows 2t-1 .. 2t+2) holds the value
        foot = torch.zeros(H, W, dtype=torch.bool, device="cuda")
        for ty in range((H + 1) // 2):
            for tx in range((W + 1) // 2):
                if 2 * ty - 1 <= py <= 2 * ty + 2 and 2 * tx - 1 <= px <= 2 * tx + 2:
                    foot[2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = True
        assert bool((foot | ~touched).all())
        touched = foot
    for b in (1, 2):
        assert bool(bad_ref[b].any())
        assert bool((bad[b] | ~bad_ref[b]).all()), (name, b)            # everything fp32 poisons is poisoned
        assert not bool((bad[b].any(-1) & ~touched).any()), (name, b)   # nothing outside the receptive field is
        if not uses_wino and (b == 1 or precision == 0):
            # NaN: exactly fp32's set in every mode.  Inf: exactly in the fp32-input mode; the split-operand modes turn the WHOLE
            # receptive field into NaN (Inf = h + l has l = Inf - Inf), where fp32 has +-Inf and ReLU maps -Inf to a finite 0
            assert torch.equal(bad[b], bad_ref[b]), (name, b, int(bad[b].sum()), int(bad_ref[b].sum()))
        out = ~touched
        assert torch.equal(y[b][out], clean[b][out])  # every output outside the receptive field keeps its bits
    a = getattr(y, "_a3d_amax", None)
    if a is not None:
        yy = y.clone()
        yy[~torch.isfinite(yy)] = 0
        assert torch.equal(a, yy.abs().flatten(1).amax(1))


def test_non_finite_candidates_are_filtered_like_detectron2(ops, oracle):
    """NaN / Inf objectness logits and deltas at a few anchors: the selection kernel and the oracle's restatement of
    find_top_rpn_proposals (valid_mask = isfinite(boxes).all(1) & isfinite(scores)) keep the same proposals."""
    O = oracle
    torch.manual_seed(9)
    B, HW = 2, (480, 640)
    shapes = [(120, 160), (60, 80), (30, 40), (15, 20), (8, 10)]
    heads = [torch.randn(B, h, w, 16, device="cuda") * 0.5 for h, w in shapes]
    heads[2][0, 3, 4, 1] = float("nan")     # a NaN objectness logit: sorts first (torch.sort and the kernel's keys alike), then filtered
    heads[3][1, 5, 6, 0] = 9.0              # a top-ranked anchor ...
    heads[3][1, 5, 6, 3 + 0] = float("inf")  # ... whose dx is Inf: a non-finite box (dw / dh would be clamped to finite ones)
    heads[1][0, 7, 7, 0] = float("inf")     # +Inf objectness
    ocfg = O.OracleCfg(score_thresh=0.0)
    cell = torch.stack([O.cell_anchors(z, ocfg.anchor_ratios) for z in ocfg.anchor_sizes])
    pb, pl, _lv, _pos, pc = ops.rpn_proposals(heads, [4, 8, 16, 32, 64], cell, HW, pre_topk=1000, post_topk=1000, nms_thresh=0.7, min_size=0.0,
                                              weights=(1.0, 1.0, 1.0, 1.0), scale_clamp=math.log(1000.0 / 16))
    gl = [h[..., :3].reshape(B, -1).cpu() for h in heads]
    gd = [h[..., 3:15].reshape(B, -1, 4).cpu() for h in heads]
    oprops = O.rpn_select(gl, gd, shapes, [HW] * B, ocfg)
    for b in range(B):
        ob, osc = oprops[b]
        n = int(pc[b])
        assert n == len(ob)
        assert bool(torch.isfinite(pb[b, :n]).all()) and bool(torch.isfinite(pl[b, :n]).all())
        assert torch.equal(pl[b, :n].cpu(), osc)
        assert float((pb[b, :n].cpu() - ob).abs().max()) < 2e-3


def test_blockwise_launches_reproduce_the_single_launch(ops, monkeypatch):
    """Tensors of 4 GiB and more run as consecutive launches over blocks of images (ops.conv2d).  With the limit lowered to a few MB
    the same code path runs on small tensors: outputs AND recorded maxima equal the single launch bit for bit -- direct, Winograd,
    residual, the two-source / four-phase upsampled conv, a row-count read from the device."""
    torch.manual_seed(3)
    B = 7
    x = torch.randn(B, 30, 40, 256, device="cuda") * torch.logspace(-2, 2, B, device="cuda")[:, None, None, None]
    res = torch.randn(B, 30, 40, 256, device="cuda")
    cases = []
    cases.append((ops.pack_conv(torch.randn(256, 256, 1, 1) / 16, torch.randn(256) * 0.1, None, 1, 0, ops.ACT_RELU), dict(res=res)))
    cases.append((ops.pack_conv(torch.randn(256, 256, 3, 3) / 48, None, None, 1, 1, ops.ACT_RELU), dict()))
    cases.append((ops.pack_conv(torch.randn(128, 256, 3, 3) / 48, None, None, 2, 1, ops.ACT_NONE), dict()))
    for pk, kw in cases:
        one = ops.conv2d(x, pk, **kw)
        monkeypatch.setattr(ops, "_ADDR_LIMIT", 3 * 30 * 40 * 256 * 4)  # three images per launch at most
        blk = ops.conv2d(x, pk, **kw)
        monkeypatch.undo()
        assert torch.equal(one, blk) and torch.equal(one._a3d_amax, blk._a3d_amax)
    # upsampled conv over a channel concat: four phase launches sharing one output and its maxima
    phases = ops.pack_conv_ups_phases(torch.randn(64, 512, 3, 3) / 68, torch.randn(64) * 0.1, None, ops.ACT_LEAKY)
    x2 = torch.randn(B, 30, 40, 256, device="cuda")
    one = ops.conv2d_ups(x, phases, x2=x2)
    monkeypatch.setattr(ops, "_ADDR_LIMIT", 2 * 30 * 40 * 256 * 4)  # two images per launch
    blk = ops.conv2d_ups(x, phases, x2=x2)
    monkeypatch.undo()
    assert torch.equal(one, blk) and torch.equal(one._a3d_amax, blk._a3d_amax)
    # linear rows with a device-side live-row count
    rows = torch.randn(900, 1024, device="cuda")
    pl = ops.pack_linear(torch.randn(256, 1024) / 32, torch.randn(256) * 0.1, None, ops.ACT_RELU)
    m_dev = torch.tensor([700], device="cuda", dtype=torch.int32)
    one = ops.linear(rows, pl, m_dev=m_dev)
    monkeypatch.setattr(ops, "_ADDR_LIMIT", 256 * 1024 * 4)
    blk = ops.linear(rows, pl, m_dev=m_dev)
    monkeypatch.undo()
    assert torch.equal(one[:700], blk[:700])


def test_batch_past_the_32_bit_limit_equals_its_halves(hip_model, oracle):
    """96 frames x 1000 proposals: the box head's fc1 input alone is 4.8 GB (the advisor's round-2 finding: such batches raised).
    The records of the whole batch equal those of its three 32-frame thirds."""
    model = hip_model
    model.roi_heads.box_predictor.test_score_thresh = 0.5
    try:
        frames = torch.from_numpy(oracle.synthetic_frames(96, seed=77)).cuda()
        whole = model.inference_batched(frames, want_masks=False)
        parts = [model.inference_batched(frames[i:i + 32].contiguous(), want_masks=False) for i in (0, 32, 64)]
        torch.cuda.synchronize()
        assert torch.equal(whole.rec_count, torch.cat([p.rec_count for p in parts]))
        assert torch.equal(whole.records, torch.cat([p.records for p in parts]))
        assert torch.equal(whole.depth, torch.cat([p.depth for p in parts]))
    finally:
        model.roi_heads.box_predictor.test_score_thresh = 0.0


def test_absmax_rows_any_row_count_length_and_alignment(ops):
    """a3d_absmax_rows (the maxima of tensors no kernel of the library produced): more rows than a grid dimension holds, row lengths
    that are not multiples of four, rows that start off a 16-byte boundary, non-finite members ignored."""
    from articulation3d_amd import _lib

    for rows, n in ((70000, 12), (5, 1023), (3, 7), (130, 4096)):
        x = torch.randn(rows * n + 1, device="cuda")[1:].view(rows, n)  # (starts 4 bytes off a 16-byte boundary)
        x[0, 0] = float("nan")
        x[-1, -1] = float("inf")
        out = torch.zeros(rows, device="cuda")
        _lib.check(_lib.lib().a3d_absmax_rows(x.data_ptr(), out.data_ptr(), rows, n, torch.cuda.current_stream().cuda_stream), "a3d_absmax_rows")
        xx = x.clone()
        xx[~torch.isfinite(xx)] = 0
        assert torch.equal(out, xx.abs().amax(1)), (rows, n)
