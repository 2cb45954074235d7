"""GPU suite (-m gpu), round 4: activations pre-split by their producer (a3d_conv_desc.x_h2) and the dual-DMA forms that consume them.

Everything here is an EQUALITY: the pre-split planes are the bits the fp16x2 loaders compute from the fp32 tensor, the kernels that
take them keep the per-output operation order of the kernels that split on the fly, so outputs must agree bit for bit
(pkg/modeling/roi_heads/roi_heads.py:185-187 box pooler -> fc1; pkg/modeling/depth_net/depth_head.py:40-46 decoder convs)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from articulation3d_amd import ops as o

    if o.DEFAULT_PRECISION != 3:
        pytest.skip("pre-split activations belong to the default (fp16x2) arithmetic")
    return o


def ref_split(x: torch.Tensor, amax: torch.Tensor) -> torch.Tensor:
    """x [B, ..., C] fp32, amax [B] -> [B, ..., C/16, 2, 16] fp16: h = fp16(x s), l = fp16(x s - h), s = 2^(14 - ilogb(amax)) (1 where amax
    is 0): conv_common.h a3d_pow2_scale + conv_bf16x3_wide.hip wx_split2h, in torch's IEEE arithmetic."""
    e = torch.frexp(amax)[1] - 1  # ilogb
    s = torch.where(amax > 0, torch.ldexp(torch.ones_like(amax), (14 - e).clamp(max=126)), torch.ones_like(amax))
    xs = x * s.view(-1, *([1] * (x.dim() - 1)))
    h = xs.half()
    l = (xs - h.float()).half()
    B = x.shape[0]
    hh = h.reshape(*x.shape[:-1], x.shape[-1] // 16, 1, 16)
    ll = l.reshape(*x.shape[:-1], x.shape[-1] // 16, 1, 16)
    return torch.cat([hh, ll], dim=-2)


def test_presplit_pass_writes_the_loader_split(ops):
    torch.manual_seed(3)
    x = torch.randn(5, 9, 7, 64, device="cuda") * torch.tensor([1.0, 1e-3, 1e4, 3.0, 0.0], device="cuda").view(5, 1, 1, 1)
    x[0, 0, 0, :8] = torch.tensor([0.0, -0.0, 1e-30, -1e-30, 6e-8, 1.0, -1.0, 0.5], device="cuda")  # zeros, values below the window
    h2 = ops.presplit_f16x2(x)
    assert h2.dtype == torch.float16 and tuple(h2.shape) == (5, 9, 7, 4, 2, 16)
    ref = ref_split(x, ops.amax_of(x))
    assert torch.equal(h2.view(torch.int16), ref.view(torch.int16))
    # two sources of a channel concat share max(amax, amax2)
    x2 = torch.randn(5, 9, 7, 32, device="cuda") * 7
    a, b = ops.presplit_f16x2(x, x2)
    am = torch.maximum(ops.amax_of(x), ops.amax_of(x2))
    assert torch.equal(a.view(torch.int16), ref_split(x, am).view(torch.int16))
    assert torch.equal(b.view(torch.int16), ref_split(x2, am).view(torch.int16))


@pytest.mark.parametrize("M,K,N", [(700, 4096, 1024), (1000, 12544, 1024), (257, 512, 320), (33, 256, 1024)], ids=lambda v: str(v))
def test_dual_dma_linear_equals_the_register_staged_kernels(ops, M, K, N):
    """conv_h2w_kernel xd (both operands by LDS-DMA, 4-stage ring) against the wide kernel that splits in its loader (tune 9) and the
    dispatcher's choice: one image per row, rows with very different maxima, ragged M and N tiles."""
    torch.manual_seed(M)
    x = torch.randn(M, K, device="cuda") * torch.logspace(-3, 3, M, device="cuda").view(M, 1)
    w = torch.randn(N, K) / K ** 0.5
    pk = ops.pack_linear(w, torch.randn(N) * 0.1, act=ops.ACT_RELU)
    xv = x.view(M, 1, 1, K)
    y_wide = ops.conv2d(xv, pk, tune=9, precision=3)
    assert ops.last_conv_variant() == "conv_h2w_kernel", ops.last_conv_variant()
    y_auto = ops.conv2d(xv, pk)
    x_h2 = ops.presplit_f16x2(xv)
    y_dd = ops.conv2d(x_h2, pk)
    assert ops.last_conv_variant() == "conv_h2w_kernel xd", ops.last_conv_variant()
    assert torch.equal(y_dd, y_wide) and torch.equal(y_dd, y_auto)
    assert torch.equal(ops.amax_of(y_dd), ops.amax_of(y_wide))
    y_lin = ops.linear(x_h2, pk)
    assert torch.equal(y_lin, y_dd.view(M, pk.cols))
    ref = torch.relu(x.double().cpu() @ w.double().t() + pk.shift[:N].double().cpu())
    err = ((y_dd.view(M, -1)[:, :N].double().cpu() - ref).abs().amax(1) / ref.abs().amax(1).clamp_min(1e-30)).max().item()
    assert err < 5e-6, err


@pytest.mark.parametrize("case", [
    dict(B=3, H=30, W=40, Cin=128, Cin2=0, Cout=256, k=3, s=1, p=1, res=False),
    dict(B=2, H=31, W=39, Cin=128, Cin2=128, Cout=128, k=3, s=1, p=1, res=False),  # channel concat, ragged tiles, borders
    dict(B=4, H=30, W=40, Cin=256, Cin2=0, Cout=512, k=1, s=2, p=0, res=False),
    dict(B=2, H=24, W=40, Cin=64, Cin2=0, Cout=256, k=1, s=1, p=0, res=True),
], ids=lambda c: f"{c['Cin']}+{c['Cin2']}to{c['Cout']}k{c['k']}s{c['s']}")
def test_dual_dma_conv_equals_the_register_staged_kernels(ops, case):
    c = case
    torch.manual_seed(17)
    scale = torch.logspace(-2, 2, c["B"], device="cuda").view(-1, 1, 1, 1)
    x = torch.randn(c["B"], c["H"], c["W"], c["Cin"], device="cuda") * scale
    x2 = torch.randn(c["B"], c["H"], c["W"], c["Cin2"], device="cuda") * scale * 3 if c["Cin2"] else None
    w = torch.randn(c["Cout"], c["Cin"] + c["Cin2"], c["k"], c["k"]) / ((c["Cin"] + c["Cin2"]) * c["k"] ** 2) ** 0.5
    pk = ops.pack_conv(w, torch.randn(c["Cout"]) * 0.1, None, c["s"], c["p"], ops.ACT_RELU)
    Ho = (c["H"] + 2 * c["p"] - c["k"]) // c["s"] + 1
    Wo = (c["W"] + 2 * c["p"] - c["k"]) // c["s"] + 1
    res = torch.randn(c["B"], Ho, Wo, pk.cols, device="cuda") if c["res"] else None
    y_ref = ops.conv2d(x, pk, x2=x2, res=res, wino=False, precision=3)  # the dispatcher's direct kernel (narrow or wide)
    assert ops.last_conv_variant().startswith(("conv_h2_kernel", "conv_h2w_kernel", "conv_h2xs", "conv_h2sg")), ops.last_conv_variant()
    if x2 is None:
        y_dd = ops.conv2d(ops.presplit_f16x2(x), pk, res=res)
    else:
        a, b = ops.presplit_f16x2(x, x2)
        y_dd = ops.conv2d(a, pk, x2=b, res=res)
    assert ops.last_conv_variant() == "conv_h2w_kernel xd", ops.last_conv_variant()
    assert torch.equal(y_dd, y_ref)
    assert torch.equal(ops.amax_of(y_dd), ops.amax_of(y_ref))


def _pyramid(B, scale=1.0):
    torch.manual_seed(23)
    return [torch.randn(B, 480 // s, 640 // s, 256, device="cuda") * scale for s in (4, 8, 16, 32)]


def _boxes(B, R, seed=0):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(B, R, 2, generator=g) * torch.tensor([600.0, 440.0])
    wh = torch.exp(torch.rand(B, R, 2, generator=g) * 5.0) + 2.0  # 3 .. 150 px: every pyramid level, sampling grids up to 3 x 3 and beyond
    b = torch.cat([xy, xy + wh], -1)
    b[0, 0] = torch.tensor([-20.0, -30.0, 700.0, 520.0])  # past the image on every side: the large-lattice walk
    b[0, 1] = torch.tensor([100.0, 100.0, 100.5, 100.5])  # sub-pixel box
    return b.cuda()


@pytest.mark.parametrize("compact", [False, True])
def test_presplit_pooler_rows_are_the_split_of_the_fp32_rows(ops, compact):
    """a3d_roialign_desc.out_h2: same pooled values (the fp32 kernel's arithmetic, bin for bin), same per-ROI maxima, and the planes are
    ref_split of the fp32 rows under the ROI's own scale; dead slots are not touched."""
    B, R = 3, 40
    feats = _pyramid(B)
    feats[1][1] *= 1e-4  # a faint image
    boxes = _boxes(B, R)
    count = torch.tensor([R, 17, 0], device="cuda", dtype=torch.int32)
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    kw = {}
    rows = B * R
    if compact:
        off = ops.count_offsets(count, R)
        rows = int(off[-1])
        kw = dict(row_offset=off, rows=rows)
    # (the pre-split pooler walks bin by bin: its rows are the split of THAT walk's fp32 rows; the default fp32 pooler's rolling-window walk
    # of round 6 sums in another order -- equal to fp32 rounding, tests/test_known_answers.py)
    saved, ops.ROI_ROLLING = ops.ROI_ROLLING, False
    try:
        f32 = ops.roi_align_fpn(feats, scales, boxes, count, 7, 0, True, zero=True, **kw)
    finally:
        ops.ROI_ROLLING = saved
    h2 = ops.roi_align_fpn(feats, scales, boxes, count, 7, 0, True, zero=True, presplit=True, **kw)
    assert h2.dtype == torch.float16 and tuple(h2.shape) == (rows, 1, 1, 49 * 16, 2, 16)
    live = torch.zeros(rows, dtype=torch.bool, device="cuda")
    for b in range(B):
        base = int(off[b]) if compact else b * R
        live[base:base + int(count[b])] = True
    am_f, am_h = ops.amax_of(f32), ops.amax_of(h2)
    assert torch.equal(am_f[live], am_h[live])
    ref = ref_split(f32.view(rows, 49 * 256), am_f)
    assert torch.equal(h2.view(rows, -1).view(torch.int16)[live], ref.view(rows, -1).view(torch.int16)[live])
    assert not h2.view(rows, -1)[~live].any()  # (zero-initialised here; the kernel left the dead rows alone)


def test_box_head_on_presplit_rows_equals_the_fp32_rows(ops):
    """Pooler -> fc1 -> fc2 as the detector runs them (roi_heads.box_batched): pre-split rows through the dual-DMA fc1 against fp32 rows
    through the register-staged fc1."""
    B, R = 2, 1000
    feats = _pyramid(B)
    boxes = _boxes(B, R, seed=5)
    count = torch.tensor([R, 640], device="cuda", dtype=torch.int32)
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    torch.manual_seed(9)
    fc1 = ops.pack_linear(torch.randn(1024, 12544) / 112.0, torch.randn(1024) * 0.1, chw=(256, 7, 7), act=ops.ACT_RELU)
    saved, ops.ROI_ROLLING = ops.ROI_ROLLING, False  # (the pre-split pooler's walk: bin by bin, see the test above)
    try:
        f32 = ops.roi_align_fpn(feats, scales, boxes, count, 7, 0, True, zero=True)
    finally:
        ops.ROI_ROLLING = saved
    h2 = ops.roi_align_fpn(feats, scales, boxes, count, 7, 0, True, zero=True, presplit=True)
    y0 = ops.linear(ops.keep_amax(f32.view(B * R, -1), f32), fc1)
    y1 = ops.linear(h2, fc1)
    assert ops.last_conv_variant() == "conv_h2w_kernel xd", ops.last_conv_variant()
    live = torch.zeros(B * R, dtype=torch.bool, device="cuda")
    live[:R] = True
    live[R:R + 640] = True
    assert torch.equal(y0[live], y1[live])


@pytest.mark.parametrize("case", [
    dict(B=2, H=16, W=64, C1=128, C2=128, Cout=64, tune=16),   # whole tiles, two sources (the decoder's last stage in miniature)
    dict(B=3, H=24, W=96, C1=128, C2=0, Cout=128, tune=16),    # one source, two column tiles
    dict(B=2, H=13, W=42, C1=64, C2=64, Cout=64, tune=16),     # partial tiles in both directions, map borders inside a tile
    dict(B=1, H=8, W=32, C1=32, C2=0, Cout=32, tune=16),       # a single tile, a single pair of chunks, 128 GEMM columns of 256
    dict(B=2, H=120, W=160, C1=128, C2=128, Cout=64, tune=0),  # the dispatcher's own choice on the decoder's 120 x 160 stage
], ids=lambda c: f"{c['B']}x{c['H']}x{c['W']}x({c['C1']}+{c['C2']})to{c['Cout']}")
def test_patch_resident_upsampled_conv_matches_the_tap_outer_form(ops, case):
    """csrc/conv_ph4p.hip (round 4): the fused four-phase upsampled conv with the input patch resident in LDS.  Same products, same term
    order per k, but the reduction runs (chunk, tap) instead of (tap, chunk): it must agree with the tap-outer form to fp32 rounding
    (2e-6 of the image's output scale; both are held to float64 at 5e-6), record the same per-image maxima to that precision, and the
    dispatcher's choice must not depend on the batch (one frame alone == the same frame inside a batch, bit for bit)."""
    import torch.nn.functional as F

    c = case
    torch.manual_seed(41)
    spread = torch.logspace(-2, 2, c["B"], device="cuda").view(-1, 1, 1, 1)
    x = torch.randn(c["B"], c["H"], c["W"], c["C1"], device="cuda") * spread
    x2 = torch.randn(c["B"], c["H"], c["W"], c["C2"], device="cuda") * spread if c["C2"] else None
    w = torch.randn(c["Cout"], c["C1"] + c["C2"], 3, 3) / (3 * (c["C1"] + c["C2"]) ** 0.5)
    bn = (torch.rand(c["Cout"]) + 0.5, torch.randn(c["Cout"]) * 0.1, torch.randn(c["Cout"]) * 0.1, torch.rand(c["Cout"]) + 0.5, 1e-3)
    phases = ops.pack_conv_ups_phases(w, None, bn, ops.ACT_RELU)
    y_old = ops.conv2d_ups(x, phases, x2=x2, tune=15)
    assert ops.last_conv_variant() == "conv_h2w_kernel ph4", ops.last_conv_variant()
    y_new = ops.conv2d_ups(x, phases, x2=x2, tune=c["tune"])
    assert ops.last_conv_variant().startswith("conv_ph4p_kernel<"), ops.last_conv_variant()
    scale = y_old.abs().flatten(1).amax(1).clamp_min(1e-30).view(-1, 1, 1, 1)
    assert float(((y_new - y_old).abs() / scale).max()) < 2e-6
    assert torch.allclose(ops.amax_of(y_new), ops.amax_of(y_old), rtol=2e-6, atol=0)
    xi = x if x2 is None else torch.cat([x, x2], -1)
    up = F.interpolate(xi[:1].double().permute(0, 3, 1, 2).cpu(), scale_factor=2, mode="nearest")
    ref = F.conv2d(up, w.double(), None, padding=1)
    sc = bn[0].double() / torch.sqrt(bn[3].double() + bn[4])
    ref = torch.relu(ref * sc.view(1, -1, 1, 1) + (bn[1].double() - bn[2].double() * sc).view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    assert float((y_new[:1, ..., : c["Cout"]].double().cpu() - ref).abs().max() / ref.abs().max()) < 5e-6
    # batch invariance of whatever the dispatcher picks (tune 0): the last frame alone
    alone = ops.conv2d_ups(x[-1:].contiguous(), phases, x2=None if x2 is None else x2[-1:].contiguous())
    batch = ops.conv2d_ups(x, phases, x2=x2)
    assert torch.equal(alone[0], batch[-1])


@pytest.mark.parametrize("shape", [(2, 30, 40), (3, 61, 79), (1, 8, 10), (2, 120, 160)])
def test_tap_product_epilogue_equals_the_two_layers(ops, shape):
    """Round 4: deconv5 + depth_pred of the depth head (pkg/modeling/depth_net/depth_head.py:51,88) without the 64-channel tensor between
    them -- the four-phase launch stores the nine tap products of every output pixel (a3d_conv_desc.dot_w / dot_y), a3d_tapsum9 adds the
    shifted planes.  Another summation order of the same 576 products per pixel: both forms against a float64 3x3 convolution of the
    SAME 64-channel tensor, the same error; results do not depend on the batch."""
    if ops.DEFAULT_PRECISION != 3:
        pytest.skip("the fused four-phase form belongs to the fp16x2 arithmetic")
    B, H, W = shape
    torch.manual_seed(B * 1000 + H)
    a = torch.randn(B, H, W, 128, device="cuda")
    c2 = torch.randn(B, H, W, 128, device="cuda")
    phases = ops.pack_conv_ups_phases(torch.randn(64, 256, 3, 3) / (3 * 256 ** 0.5), torch.randn(64) * 0.1, None, ops.ACT_RELU)
    w9 = (torch.randn(3, 3, 64) / 24).cuda()
    x = ops.conv2d_ups(a, phases, x2=c2)
    two = ops.conv3x3_to1(x, w9, 0.3)
    one = ops.conv2d_ups_to1(a, phases, w9, 0.3, x2=c2)
    assert one is not None and ops.last_conv_variant().endswith(" dot"), ops.last_conv_variant()
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w9.double().permute(2, 0, 1)[None], torch.tensor([0.3], dtype=torch.float64, device="cuda"), 1, 1)[:, 0]
    e2, e1 = (float((o.double() - ref).abs().max() / ref.abs().max()) for o in (two, one))
    assert e1 < 1e-6 and e2 < 1e-6 and e1 < 3 * e2 + 1e-7, (e1, e2)
    if B > 1:  # batch invariance, bit for bit
        alone = ops.conv2d_ups_to1(a[1:2].contiguous(), phases, w9, 0.3, x2=c2[1:2].contiguous())
        assert torch.equal(alone[0], one[1])
        # a batch past the 32-bit offsets of one launch runs as blocks of images through the SAME two kernels (advisor, round 4: it used to
        # switch to the two-layer form, another summation order): shrink the limit so that this batch needs one launch per image
        limit = ops._ADDR_LIMIT
        try:
            ops._ADDR_LIMIT = max(9 * 4 * H * W * 4, H * W * 128 * 4)
            blocks = ops.conv2d_ups_to1(a, phases, w9, 0.3, x2=c2)
        finally:
            ops._ADDR_LIMIT = limit
        assert blocks is not None and torch.equal(blocks, one)
