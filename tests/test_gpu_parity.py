"""GPU suite (-m gpu): the HIP path, called through the C ABI, against the CPU oracle.

Methodology.  Continuous stages (convolutions, ROIAlign, heads, depth) are compared END TO END with the
oracle at a stated fp32 tolerance.  Discrete stages (top-k, NMS keep masks, detection indices, pasted masks)
are discontinuous in their inputs, so each is compared on IDENTICAL inputs -- the oracle stage is fed the
very tensors the HIP stage consumed -- and must then agree BIT-EXACTLY.  Box coordinates that pass through
`exp` are compared at 1e-3 px (device expf vs libm differ by an ulp).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import golden_inputs as G

pytestmark = pytest.mark.gpu
HW = (480, 640)
NAMES = ("p2", "p3", "p4", "p5", "p6")


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from articulation3d_amd import ops as o

    return o


# ------------------------------------------------------------------------------------------ conv-GEMM units
CONV_CASES = [
    dict(B=2, H=24, W=40, Cin=64, Cout=256, k=1, s=1, p=0),
    dict(B=2, H=24, W=40, Cin=256, Cout=64, k=1, s=1, p=0),
    dict(B=2, H=24, W=40, Cin=64, Cout=64, k=3, s=1, p=1),
    dict(B=3, H=15, W=20, Cin=256, Cout=256, k=3, s=1, p=1),
    dict(B=2, H=24, W=40, Cin=256, Cout=512, k=1, s=2, p=0),
    dict(B=2, H=8, W=10, Cin=256, Cout=15, k=1, s=1, p=0),
    dict(B=1, H=7, W=9, Cin=32, Cout=36, k=3, s=2, p=1),  # ragged M and N tiles, stride 2 with padding
    dict(B=1, H=1, W=1, Cin=2048, Cout=512, k=1, s=1, p=0),  # single output row
]


@pytest.mark.parametrize("c", CONV_CASES, ids=lambda c: f"{c['Cin']}to{c['Cout']}k{c['k']}s{c['s']}")
def test_conv_gemm_vs_torch_fp32(ops, c):
    torch.manual_seed(1)
    x = torch.randn(c["B"], c["Cin"], c["H"], c["W"])
    w = torch.randn(c["Cout"], c["Cin"], c["k"], c["k"]) / (c["Cin"] * c["k"] ** 2) ** 0.5
    b = torch.randn(c["Cout"])
    ref = F.relu(F.conv2d(x, w, b, stride=c["s"], padding=c["p"]))
    p = ops.pack_conv(w, b, None, c["s"], c["p"], ops.ACT_RELU, device="cuda")
    y = ops.conv2d(nhwc(x).cuda(), p)
    assert rel(y[..., : c["Cout"]].permute(0, 3, 1, 2), ref) < 5e-6  # fp32 MFMA vs fp32 oneDNN


@pytest.mark.parametrize("shape", [(3, 60, 80, 64, 64), (2, 120, 160, 128, 256), (2, 61, 79, 32, 36), (5, 60, 80, 256, 96)],
                         ids=lambda s: "x".join(map(str, s)))
def test_one_launch_winograd_is_bit_identical_to_the_two_launch_form(ops, shape):
    """csrc/conv_wino_fused.hip (input transform inside the GEMM loader, 2-D tile blocks staged by LDS-DMA, 16 plane accumulators)
    runs the same k order and the same transform expressions as wino_input_kernel + wino_gemm_kernel: equal bits, incl. odd image
    sizes (partial tiles / blocks), Cout that is not a multiple of 64, every block shape, ReLU gate and LeakyReLU epilogues."""
    B, H, W, Cin, Cout = shape
    torch.manual_seed(5)
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5)
    bn = (torch.rand(Cout) + 0.5, torch.randn(Cout) * 0.1, torch.randn(Cout) * 0.1, torch.rand(Cout) + 0.5, 1e-5)
    pk = ops.pack_conv(w, None, bn, 1, 1, ops.ACT_LEAKY)
    y1 = ops.conv2d(x, pk, precision=0)
    assert ops.last_conv_variant().startswith("wino_fused_kernel"), ops.last_conv_variant()
    y2 = ops.conv2d(x, pk, tune=7, precision=0)
    assert ops.last_conv_variant().startswith("wino_gemm_kernel"), ops.last_conv_variant()
    assert torch.equal(y1, y2)
    ref = F.leaky_relu(F.batch_norm(F.conv2d(x.permute(0, 3, 1, 2).cpu(), w, padding=1), bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5), 0.01)
    assert rel(y1[..., :Cout].permute(0, 3, 1, 2), ref) < 5e-6
    gate = torch.randn(B, H, W, pk.cols, device="cuda")
    pk.act = ops.ACT_RELU
    assert torch.equal(ops.conv2d(x, pk, gate=gate, precision=0), ops.conv2d(x, pk, gate=gate, tune=7, precision=0))


@pytest.mark.parametrize("precision", [0, 2])
def test_shared_winograd_input_transform_is_bit_identical(ops, precision):
    """ops.share_wino_input: two 3x3 layers reading one tensor (RPN conv + depth lateral conv per FPN level; plane + axis head
    conv1) run ONE wino_input_kernel; V depends on the input only, so each layer's output equals its unshared output bit for bit."""
    torch.manual_seed(11)
    x = torch.randn(3, 60, 80, 256, device="cuda")
    other = torch.randn(3, 60, 80, 256, device="cuda")
    pks = [ops.pack_conv(torch.randn(co, 256, 3, 3) / 48, torch.randn(co) * 0.1, None, 1, 1, ops.ACT_RELU) for co in (256, 96)]
    alone = [ops.conv2d(x, pk, precision=precision) for pk in pks]
    two_launch = [ops.conv2d(x, pk, precision=precision, tune=7 if precision == 0 else 0) for pk in pks]
    ops.CONV_TIMING = []
    try:
        with ops.share_wino_input([x]):
            shared = [ops.conv2d(x, pk, precision=precision) for pk in pks]
            unlisted = ops.conv2d(other, pks[0], precision=precision)  # a tensor that is not in the scope is untouched by it
        torch.cuda.synchronize()
        names = [t[0] for t in ops.CONV_TIMING]
    finally:
        ops.CONV_TIMING = None
    for a, b, c in zip(alone, two_launch, shared):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert torch.equal(unlisted, ops.conv2d(other, pks[0], precision=precision))
    gemm = "wino_gemm_x3" if precision == 2 else "wino_gemm_kernel"  # (bf16x3: the narrow or a wide split-operand GEMM, by problem size)
    assert sum(n.startswith(gemm) for n in names) >= 2


@pytest.mark.parametrize("shape", [(64, 60, 80, 256, 256), (330, 30, 40, 128, 128), (2100, 14, 14, 256, 256), (21, 121, 159, 64, 128)],
                         ids=lambda s: "x".join(map(str, s)))
def test_wide_split_operand_winograd_gemm_is_bit_identical_to_the_64_wide_form(ops, shape):
    """wino_gemm_x3w_kernel<4> (128 tiles x 128 channels, pre-split weight planes global -> LDS by LDS-DMA, 512 threads) keeps the
    per-output operation order of wino_gemm_x3_kernel: equal bits, so the launcher may choose between them by problem size
    (ragged tile / channel edges included)."""
    B, H, W, Cin, Cout = shape
    torch.manual_seed(3)
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5)
    pk = ops.pack_conv(w, torch.randn(Cout) * 0.1, None, 1, 1, ops.ACT_LEAKY)
    y_wide = ops.conv2d(x, pk, precision=2)
    assert ops.last_conv_variant().startswith("wino_gemm_x3w_kernel<4>"), ops.last_conv_variant()
    y_64 = ops.conv2d(x, pk, precision=2, tune=8)
    assert ops.last_conv_variant() == "wino_gemm_x3_kernel", ops.last_conv_variant()
    assert torch.equal(y_wide, y_64)
    ref = F.leaky_relu(F.conv2d(x[-2:].permute(0, 3, 1, 2).double().cpu(), w.double(), pk.shift[:Cout].double().cpu(), padding=1), 0.01)
    assert rel(y_wide[-2:, :, :, :Cout].permute(0, 3, 1, 2).double(), ref) < 2e-6  # (the last images: the ragged tile block is there)


@pytest.mark.parametrize("shape", [(64, 30, 40, 256, 256), (3, 61, 79, 256, 384), (70, 14, 14, 256, 256), (5, 120, 160, 256, 128), (20, 15, 20, 512, 512)],
                         ids=lambda s: "x".join(map(str, s)))
def test_ping_pong_winograd_gemm_keeps_the_bits_of_the_lockstep_and_64_tile_forms(ops, shape):
    """Round 5: the fp16x2 Winograd GEMM (the 3x3 layers of the FPN, RPN, depth laterals and ROI heads: planercnn.py:150,168) runs its
    512-thread workgroup as two halves in antiphase -- waves 4-7 half a chunk behind waves 0-3, each wave alternating a memory phase
    (a chunk's 12 fragment reads + its 4 DMA pieces) with a compute phase (its 12 MFMAs) -- on every problem size.  Planes, chunks,
    steps and product terms per accumulator are those of the lockstep loop (tune 23) and of the 64-tile form (tune 24): equal bits, so
    which form runs is free (ragged tile / channel blocks and a 16-chunk plane included)."""
    if ops.DEFAULT_PRECISION != 3:
        pytest.skip("the fp16x2 arithmetic's kernel")
    B, H, W, Cin, Cout = shape
    torch.manual_seed(11)
    x = torch.randn(B, H, W, Cin, device="cuda") * torch.logspace(-2, 2, B, device="cuda").view(B, 1, 1, 1)  # (per-image scales differ)
    w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5)
    pk = ops.pack_conv(w, torch.randn(Cout) * 0.1, None, 1, 1, ops.ACT_RELU)
    y = ops.conv2d(x, pk, precision=3)
    assert ops.last_conv_variant() == "wino_gemm_h2w_kernel<4>", ops.last_conv_variant()
    lock = ops.conv2d(x, pk, precision=3, tune=23)
    assert ops.last_conv_variant() == "wino_gemm_h2w_kernel<4>", ops.last_conv_variant()
    narrow = ops.conv2d(x, pk, precision=3, tune=24)
    assert ops.last_conv_variant() == "wino_gemm_h2w_kernel<2>", ops.last_conv_variant()
    assert torch.equal(y, lock) and torch.equal(y, narrow)
    ref = F.relu(F.conv2d(x[-2:].permute(0, 3, 1, 2).double().cpu(), w.double(), pk.shift[:Cout].double().cpu(), padding=1))
    assert rel(y[-2:, :, :, :Cout].permute(0, 3, 1, 2).double(), ref) < 2e-6


@pytest.mark.parametrize("case", [dict(B=33000, H=1, W=1, Cin=4096, Cout=1024, k=1, s=1, res=False),  # default dispatch: deep reduction
                                  dict(B=3, H=30, W=40, Cin=256, Cout=1000, k=1, s=1, res=True),     # ragged channel tile, residual
                                  dict(B=2, H=33, W=41, Cin=64, Cout=256, k=3, s=2, res=False),      # taps, stride, ragged pixel tile
                                  dict(B=2, H=16, W=24, Cin=96, Cout=512, k=1, s=1, res=False)],     # six-chunk loop: mostly prologue / epilogue
                         ids=lambda c: f"{c['B']}x{c['H']}x{c['W']}x{c['Cin']}->{c['Cout']}k{c['k']}s{c['s']}")
def test_wide_split_operand_direct_kernel_is_bit_identical_to_the_narrow_one(ops, case):
    """conv_x3w_kernel (256 x 256 tiles, weight planes pre-split once per layer and moved global -> LDS by LDS-DMA, one barrier per
    16-deep chunk) keeps conv_x3_kernel's per-output operation order: equal bits, so the launcher may choose by problem shape."""
    c = case
    torch.manual_seed(9)
    x = torch.randn(c["B"], c["H"], c["W"], c["Cin"], device="cuda")
    w = torch.randn(c["Cout"], c["Cin"], c["k"], c["k"]) / (c["k"] * c["Cin"] ** 0.5)
    pk = ops.pack_conv(w, torch.randn(c["Cout"]) * 0.1, None, c["s"], c["k"] // 2, ops.ACT_RELU)
    Ho = (c["H"] + 2 * (c["k"] // 2) - c["k"]) // c["s"] + 1
    Wo = (c["W"] + 2 * (c["k"] // 2) - c["k"]) // c["s"] + 1
    res = torch.randn(c["B"], Ho, Wo, pk.cols, device="cuda") if c["res"] else None
    wide = ops.conv2d(x, pk, precision=2, tune=9, res=res)
    assert ops.last_conv_variant() == "conv_x3w_kernel", ops.last_conv_variant()
    narrow = ops.conv2d(x, pk, precision=2, tune=8, res=res)
    assert ops.last_conv_variant().startswith("conv_x3_kernel"), ops.last_conv_variant()
    assert torch.equal(wide, narrow)
    if c["Cin"] >= 4096:
        ops.conv2d(x, pk, precision=2, res=res)
        assert ops.last_conv_variant() == "conv_x3w_kernel", ops.last_conv_variant()
    nb = min(c["B"], 64)  # (the float64 reference on the leading images / rows only)
    ref = F.conv2d(x[:nb].permute(0, 3, 1, 2).double().cpu(), w.double(), pk.shift[:c["Cout"]].double().cpu(), stride=c["s"], padding=c["k"] // 2)
    if res is not None:
        ref = ref + res[:nb, :, :, :c["Cout"]].permute(0, 3, 1, 2).double().cpu()
    # fp32 accumulation over K terms: ~sqrt(K) * 2^-24 of the output scale (4e-6 at K = 4096)
    assert rel(wide[:nb, :, :, :c["Cout"]].permute(0, 3, 1, 2).double(), F.relu(ref)) < 5e-6


def test_split_k_head_fc_in_the_split_operand_arithmetic(ops):
    """The 50176-deep head FCs (plane_head.py:76, axis_head.py, split-K 32 by K alone) run the wide bf16x3 kernel with blockIdx.y =
    K slice and the shared reduce launch: fp32-grade against float64, and -- the order of the slices being fixed -- a row's result
    does not depend on how many rows are in the batch."""
    torch.manual_seed(21)
    K, N = 16384, 1024
    x = torch.randn(300, K, device="cuda")
    w = torch.randn(N, K) / K ** 0.5
    b = torch.randn(N) * 0.1
    pk = ops.pack_linear(w, b, None, ops.ACT_RELU)
    sk = ops.choose_splitk(300, pk.cols, K)
    assert sk == 32
    ref = F.relu(x.double().cpu() @ w.double().t() + b.double())
    for prec, name in ((3, "conv_h2w_kernel sk32"), (2, "conv_x3w_kernel sk32")):
        y = ops.linear(x, pk, splitk=sk, precision=prec)
        assert ops.last_conv_variant() == name, ops.last_conv_variant()
        assert rel(y[:, :N].double(), ref) < 5e-6
        few = ops.linear(x[:7].contiguous(), pk, splitk=sk, precision=prec)
        assert torch.equal(few, y[:7])
    y0 = ops.linear(x, pk, splitk=sk, precision=0)  # the fp32-input MFMA split-K kernel: same slices, same order
    assert rel(y0[:, :N].double(), ref) < 5e-6


H2_CASES = [  # B, H, W, Cin, Cout, k, stride, kwargs, expected kernel
    (6, 30, 40, 256, 1024, 1, 1, {}, "conv_h2_kernel"),
    (5, 30, 40, 1024, 256, 1, 1, {}, "conv_h2_kernel"),
    (4, 61, 79, 128, 128, 3, 2, {}, "conv_h2_kernel"),           # taps, stride, ragged tiles
    (3, 61, 79, 64, 64, 3, 1, {}, "conv_c3p_kernel<1, 16>"),     # a 3x3 s1 layer the 128-wide Winograd tiles do not fit: direct form, patch-resident (round 4)
    (3, 61, 79, 64, 64, 3, 1, dict(tune=10), "conv_h2_kernel"),   # ... and its tap-outer form
    (700, 1, 1, 4096, 1024, 1, 1, dict(tune=9, precision=3), "conv_h2w_kernel"),
    (300, 1, 1, 16384, 1024, 1, 1, dict(splitk=32), "conv_h2w_kernel sk32"),
    (5, 30, 40, 256, 256, 3, 1, {}, "wino_gemm_h2w_kernel"),
    (3, 61, 79, 256, 384, 3, 1, {}, "wino_gemm_h2w_kernel"),     # ragged tile / channel blocks
    (3, 60, 80, 128, 128, 3, 1, {}, "conv_c3p_kernel<2, 16>"),   # under 256 input channels: the direct form (ops.conv2d's rule), patch-resident
    (3, 30, 40, 128, 128, 3, 1, {}, "conv_h2_kernel"),           # ... on a map the 8 x 32 tiles cover badly: the tap-outer kernel
    (3, 60, 80, 128, 128, 3, 1, dict(wino=True), "wino_gemm_h2w_kernel"),
    (70, 14, 14, 256, 256, 3, 1, {}, "wino_gemm_h2w_kernel"),
]


@pytest.mark.parametrize("case", H2_CASES, ids=lambda c: f"{c[0]}x{c[1]}x{c[2]}x{c[3]}->{c[4]}k{c[5]}s{c[6]}-{c[8].split('<')[0].split(' ')[0]}{'-' + '-'.join(map(str, c[7])) if c[7] else ''}")
def test_fp16x2_kernels_are_fp32_grade_batch_invariant_and_record_their_maxima(ops, case):
    """The default arithmetic (a3d_conv_desc.precision == 3): x * s = h + l in fp16 with a power-of-two scale per image, three MFMAs
    per k step.  With images 10^6 apart in magnitude in ONE batch: (1) every image's error against float64 is no larger than bf16x3's
    (x 1.25) and fp32-grade in absolute terms; (2) an image's result does not depend on the rest of the batch (bitwise); (3) the
    per-image output maxima the epilogue records for the next layer are exact."""
    B, H, W, Cin, Cout, k, st, kw, kernel = case
    torch.manual_seed(2)
    x = torch.relu(torch.randn(B, H, W, Cin, device="cuda")) * torch.exp(1.5 * torch.randn(Cin, device="cuda")) \
        * torch.logspace(-3, 3, B, device="cuda")[:, None, None, None]
    w = torch.randn(Cout, Cin, k, k) / (k * Cin ** 0.5)
    pk = ops.pack_conv(w, torch.randn(Cout) * 0.1, None, st, k // 2, ops.ACT_RELU)
    kw3 = dict(kw)
    kw3.setdefault("precision", 3)
    y = ops.conv2d(x, pk, **kw3)
    assert ops.last_conv_variant().startswith(kernel), ops.last_conv_variant()
    y2 = ops.conv2d(x, pk, **dict(kw, precision=2, wino=kernel.startswith("wino")))  # bf16x3 in the same (Winograd / direct) form
    nb = min(B, 8)
    sel = torch.linspace(0, B - 1, nb).long().cuda()  # (float64 reference on a spread of the images)
    ref = F.relu(F.conv2d(x[sel].permute(0, 3, 1, 2).double(), w.double().cuda(), pk.shift[:Cout].double(), stride=st, padding=k // 2)).permute(0, 2, 3, 1)
    err = lambda t: ((t[sel][..., :Cout].double() - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1))
    e3, e2 = err(y), err(y2)
    assert float(e3.max()) < 2e-6 and bool((e3 <= 1.25 * e2 + 1e-8).all()), (e3.tolist(), e2.tolist())
    assert torch.equal(y._a3d_amax, y.abs().flatten(1).amax(1))
    alone = ops.conv2d(x[B // 2:B // 2 + 1].contiguous(), pk, **kw3)
    assert torch.equal(alone[0], y[B // 2])


def test_fp16x2_scaling_edge_cases(ops):
    """The per-image power-of-two scale at its limits: an all-zero image (scale 1, output = the shift), images at 1e-30 and 1e+30
    (fp16's range is 6e-8 .. 6.5e4: without the scale both would vanish / overflow), and one outlier 10^5 x the rest of its image
    (the rest sits below 2^-16 of the maximum: still far inside the 2^-18 the two planes keep at full precision)."""
    torch.manual_seed(6)
    B, H, W, Cin, Cout = 5, 30, 40, 256, 256
    x = torch.randn(B, H, W, Cin, device="cuda")
    x[0] = 0.0
    x[1] *= 1e-30
    x[2] *= 1e30
    x[3, 7, 9, 11] = 3e5
    w = torch.randn(Cout, Cin, 1, 1) / 16
    pk = ops.pack_conv(w, torch.randn(Cout) * 0.1, None, 1, 0, ops.ACT_NONE)
    y = ops.conv2d(x, pk, precision=3)
    assert ops.last_conv_variant().startswith("conv_h2_kernel")
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double().cuda(), pk.shift[:Cout].double()).permute(0, 2, 3, 1)
    assert torch.equal(y[0], pk.shift[None, None, :].expand(H, W, -1))  # zero image: exactly the shift
    for b in (1, 2, 4):
        e = float((y[b].double() - ref[b]).norm() / ref[b].norm())
        assert e < 1e-6, (b, e)
    # the outlier image: every pixel but the outlier's keeps fp32-grade relative accuracy
    m = torch.ones(H, W, dtype=torch.bool, device="cuda")
    m[7, 9] = False
    e = float((y[3][m].double() - ref[3][m]).norm() / ref[3][m].norm())
    assert e < 1e-6, e
    assert bool(torch.isfinite(y).all())
    # a Winograd layer with the same extremes
    w3 = torch.randn(Cout, Cin, 3, 3) / 48
    pk3 = ops.pack_conv(w3, None, None, 1, 1, ops.ACT_NONE)
    y3 = ops.conv2d(x, pk3, precision=3)
    assert ops.last_conv_variant().startswith("wino_gemm_h2w_kernel")
    ref3 = F.conv2d(x.permute(0, 3, 1, 2).double(), w3.double().cuda(), padding=1).permute(0, 2, 3, 1)
    assert float(y3[0].abs().max()) == 0.0
    for b in (1, 2, 4):
        e = float((y3[b].double() - ref3[b]).norm() / ref3[b].norm())
        assert e < 2e-6, (b, e)
    assert bool(torch.isfinite(y3).all())


def test_fp16x2_wide_and_narrow_direct_kernels_agree_bit_for_bit(ops):
    """conv_h2w_kernel (256 x 256 tiles, pre-split scaled filter by LDS-DMA) and conv_h2_kernel keep one operation order per output, as
    their bf16x3 counterparts do: the launcher may choose by problem size."""
    torch.manual_seed(4)
    x = torch.randn(600, 1, 1, 2048, device="cuda") * torch.logspace(-2, 2, 600, device="cuda")[:, None, None, None]
    pk = ops.pack_conv(torch.randn(512, 2048, 1, 1) / 45, torch.randn(512) * 0.1, None, 1, 0, ops.ACT_RELU)
    wide = ops.conv2d(x, pk, precision=3, tune=9)
    assert ops.last_conv_variant() == "conv_h2w_kernel", ops.last_conv_variant()
    narrow = ops.conv2d(x, pk, precision=3, tune=8)
    assert ops.last_conv_variant().startswith("conv_h2_kernel"), ops.last_conv_variant()
    assert torch.equal(wide, narrow)


@pytest.mark.parametrize("case", [(8, 120, 160, 64, 256, True), (8, 120, 160, 64, 256, False), (8, 60, 80, 128, 512, True), (5, 37, 41, 128, 128, False),
                                  (4, 30, 40, 256, 1024, True), (2, 60, 80, 256, 256, False), (64, 15, 20, 256, 128, True), (1, 3, 5, 64, 128, True),
                                  (37, 1, 1, 128, 256, False)])
def test_activation_stationary_pointwise_kernel_agrees_bit_for_bit_with_the_tiled_one(ops, case):
    """conv_h2xs_kernel (conv_xs_h2.hip: a wave keeps its 32 pixels' channels in registers and walks over all of N; the filter streams
    through an LDS ring) against conv_h2_kernel on the same layer: same split, same k order, same epilogue -- outputs AND recorded
    per-image maxima are identical, so the launcher may choose by layer size.  Ragged pixel counts (waves that straddle two images,
    a last tile with rows past the end), the tail split along N and every Cin the kernel takes."""
    B, H, W, Cin, Cout, with_res = case
    torch.manual_seed(B * 1000 + Cin)
    x = torch.randn(B, H, W, Cin, device="cuda") * torch.logspace(-2, 2, B, device="cuda")[:, None, None, None]
    res = torch.randn(B, H, W, Cout, device="cuda") if with_res else None
    pk = ops.pack_conv(torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5, torch.randn(Cout) * 0.1, None, 1, 0, ops.ACT_RELU)
    a = ops.conv2d(x, pk, res=res, precision=3, tune=13)
    assert ops.last_conv_variant() == f"conv_h2xs_kernel<{Cin}>", ops.last_conv_variant()
    b = ops.conv2d(x, pk, res=res, precision=3, tune=14)
    assert ops.last_conv_variant().startswith(("conv_h2_kernel", "conv_h2w_kernel")), ops.last_conv_variant()
    assert torch.equal(a, b)
    assert torch.equal(ops.amax_of(a), ops.amax_of(b))
    assert float(ops.amax_of(a).max()) > 0


@pytest.mark.parametrize("case", [(1, 30, 40, 1024, 256, 1, "none"), (1, 15, 20, 2048, 512, 1, "none"), (1, 15, 20, 512, 2048, 1, "res"), (1, 60, 80, 512, 256, 2, "none"),
                                  (1, 30, 40, 1024, 256, 1, "ups"), (3, 13, 17, 192, 96, 1, "res"), (2, 29, 31, 64, 128, 2, "none"), (1, 1, 1, 256, 32, 1, "none"),
                                  (5, 9, 7, 320, 64, 1, "res")])
def test_small_grid_pointwise_kernel_agrees_bit_for_bit_with_the_tiled_one(ops, case):
    """conv_h2sg_kernel (conv_sg_h2.hip: one wave per 32 x 32 output tile, operand fragments straight into a register ring, no LDS, no
    barrier -- what the deep 1x1 layers of the trunk run as when ONE frame arrives, the reference's own loop, tools/inference.py:215-228)
    against conv_h2_kernel on the same layer: same split, same k order, same three products per chunk into one accumulator, same epilogue
    -- outputs AND recorded per-image maxima are identical, so the launcher chooses by grid size and a frame's bits do not depend on its
    batch.  Strides, residual rows (plain and from the 2 x coarser level), both ring depths, ragged pixel counts, waves that straddle
    images of magnitudes 10^4 apart."""
    B, H, W, Cin, Cout, stride, resk = case
    torch.manual_seed(B * 1000 + Cin + Cout)
    x = torch.relu(torch.randn(B, H, W, Cin, device="cuda")) * torch.logspace(-2, 2, B, device="cuda")[:, None, None, None]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    kw = {}
    if resk == "res":
        kw = dict(res=torch.randn(B, Ho, Wo, Cout, device="cuda"))
    elif resk == "ups":
        kw = dict(res=torch.randn(B, Ho // 2, Wo // 2, Cout, device="cuda"), res_ups=True)
    pk = ops.pack_conv(torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5, torch.randn(Cout) * 0.1, None, stride, 0, ops.ACT_RELU)
    a = ops.conv2d(x, pk, precision=3, tune=17, **kw)
    assert ops.last_conv_variant() == f"conv_h2sg_kernel<{8 if Cin % 128 == 0 else 4}>", ops.last_conv_variant()
    b = ops.conv2d(x, pk, precision=3, tune=10 if Cout <= 64 else 11, **kw)
    assert ops.last_conv_variant().startswith("conv_h2_kernel"), ops.last_conv_variant()
    assert torch.equal(a, b)
    assert torch.equal(ops.amax_of(a), ops.amax_of(b)) and torch.equal(ops.amax_of(a), a.abs().flatten(1).amax(1))
    # the launcher's rule: this form up to ~1200 tiles (one frame on the stride-16 / 32 levels, 1-4 frames on the deepest), the tiled kernels above
    c = ops.conv2d(x, pk, precision=3, **kw)
    tiles = -(-B * Ho * Wo // 32) * (Cout // 32)
    assert ops.last_conv_variant().startswith("conv_h2sg_kernel" if tiles <= 1280 else ("conv_h2_kernel", "conv_h2w_kernel", "conv_h2xs")), (tiles, ops.last_conv_variant())
    assert torch.equal(a, c)


def test_launch_plans_repeat_the_full_path_bit_for_bit(ops):
    """ops._conv2d_launch keeps the finished descriptor of a module-cached layer per (input shape, residual form, arithmetic mode) and
    writes only the tensor pointers on later calls (host time of the reference's per-frame loop).  A planned launch must be the launch
    the full path makes: same kernel, same bits, same recorded maxima -- for direct, small-grid, Winograd (workspace + plane-split
    scratch) and residual forms, in the three arithmetics; a call with any option goes the full way and leaves the plan alone."""
    torch.manual_seed(9)
    cases = [(1, 30, 40, 1024, 256, 1, 1, False), (2, 60, 80, 256, 256, 1, 1, True), (1, 30, 40, 256, 256, 3, 1, False), (3, 15, 20, 512, 512, 3, 1, False),
             (2, 61, 79, 128, 128, 3, 2, False), (1, 60, 80, 64, 64, 3, 1, False)]
    saved = ops.DEFAULT_PRECISION
    try:
        for mode in (3, 2, 0):
            ops.DEFAULT_PRECISION = mode
            for B, H, W, Cin, Cout, k, st, with_res in cases:
                x = torch.relu(torch.randn(B, H, W, Cin, device="cuda")) * torch.logspace(-1, 1, B, device="cuda")[:, None, None, None]
                Ho, Wo = (H + 2 * (k // 2) - k) // st + 1, (W + 2 * (k // 2) - k) // st + 1
                res = torch.randn(B, Ho, Wo, Cout, device="cuda") if with_res else None
                pk = ops.pack_conv(torch.randn(Cout, Cin, k, k) / (k * Cin ** 0.5), torch.randn(Cout) * 0.1, None, st, k // 2, ops.ACT_RELU)
                ops.LAUNCH_PLANS = False
                ref = ops.conv2d(x, pk, res=res)
                vref, aref = ops.last_conv_variant(), getattr(ref, "_a3d_amax", None)
                assert not getattr(pk, "_plans", None)
                ops.LAUNCH_PLANS = True
                first = ops.conv2d(x, pk, res=res)   # full path, creates the plan
                assert len(pk._plans) == 1
                again = ops.conv2d(x, pk, res=res)   # planned
                assert ops.last_conv_variant() == vref, (mode, vref, ops.last_conv_variant())
                assert torch.equal(ref, first) and torch.equal(ref, again), (mode, B, H, W, Cin, Cout, k)
                if aref is not None:
                    assert torch.equal(aref, again._a3d_amax)
                other = ops.conv2d(x[:1].contiguous(), pk, res=None if res is None else res[:1].contiguous())  # another shape: its own plan
                assert torch.equal(other[0], ref[0]) and len(pk._plans) == (2 if B > 1 else 1)
                ops.conv2d(x, pk, res=res, act=ops.ACT_NONE)  # an option: the full path, no new plan
                assert len(pk._plans) == (2 if B > 1 else 1)
    finally:
        ops.DEFAULT_PRECISION, ops.LAUNCH_PLANS = saved, True


@pytest.mark.parametrize("case", [(4, 120, 160, 64, 256, 64), (5, 119, 161, 64, 256, 64), (16, 60, 80, 128, 512, 128), (15, 59, 81, 128, 512, 128)])
def test_back_to_back_pointwise_pair_equals_its_two_launches_bit_for_bit(ops, case):
    """Round 6: conv3 + FrozenBN + residual + ReLU of a bottleneck block and conv1 + FrozenBN + ReLU of the next block
    as ONE launch (ops.conv2d_b2b, csrc/conv_xs_b2b.hip, include/a3d.h a3d_conv_b2b; BottleneckBlock reached from planercnn.py:29,150).
    The block output y and its recorded per-image maxima equal the activation-stationary kernel's launch bit for bit; the squeezed
    tensor z and its maxima equal the bf16x3 launch of the second layer ON that y bit for bit -- same exact three-way split, same chunk
    order, same six products per chunk into one accumulator, same epilogue -- without y being read back.  Both instantiations (res2's
    64 -> 256 -> 64 and res3's 128 -> 512 -> 128, which ops does not dispatch: a tie in time), ragged pixel counts (a last tile with
    rows past the end, waves that straddle two images)."""
    B, H, W, Cin, Cmid, Cout2 = case
    torch.manual_seed(B * 100 + Cin)
    x = torch.relu(torch.randn(B, H, W, Cin, device="cuda")) * torch.logspace(-1.5, 1.5, B, device="cuda")[:, None, None, None]
    res = torch.relu(torch.randn(B, H, W, Cmid, device="cuda")) * torch.logspace(-1, 1, B, device="cuda")[:, None, None, None]
    bn = lambda c: (1.0 + 0.2 * torch.randn(c), 0.1 * torch.randn(c), 0.05 * torch.randn(c), 1.0 + 0.1 * torch.rand(c), 1e-5)
    p1 = ops.pack_conv(torch.randn(Cmid, Cin, 1, 1) / Cin ** 0.5, None, bn(Cmid), 1, 0, ops.ACT_RELU)
    p2 = ops.pack_conv(torch.randn(Cout2, Cmid, 1, 1) / Cmid ** 0.5, None, bn(Cout2), 1, 0, ops.ACT_RELU)
    y0 = ops.conv2d(x, p1, res=res, precision=3, tune=13)
    assert ops.last_conv_variant() == f"conv_h2xs_kernel<{Cin}>", ops.last_conv_variant()
    z0 = ops.conv2d(y0, p2, precision=2)
    assert ops.last_conv_variant().startswith("conv_x3_kernel"), ops.last_conv_variant()
    saved = ops.B2B_PAIRS, ops.B2B_MIN_PIXELS
    ops.B2B_PAIRS, ops.B2B_MIN_PIXELS = ((64, 64), (128, 128)), 1
    try:
        pair = ops.conv2d_b2b(x, p1, res, p2)
    finally:
        ops.B2B_PAIRS, ops.B2B_MIN_PIXELS = saved
    assert pair is not None and ops.last_conv_variant() == f"conv_h2xs_b2b_kernel<{Cin},{Cout2}>", ops.last_conv_variant()
    y1, z1 = pair
    assert torch.equal(y0, y1) and torch.equal(ops.amax_of(y0), ops.amax_of(y1))
    assert torch.equal(z0, z1) and torch.equal(ops.amax_of(z0), ops.amax_of(z1))
    assert float(ops.amax_of(z1).min()) > 0 and bool(torch.isfinite(z1).all())
    # ... and against float64 on a slice: the second layer carries full 24-bit operands (bf16x3), the first the fp16x2 law
    xs, rs = x[:1, :8].double(), res[:1, :8].double()
    yd = torch.relu(torch.einsum("bhwc,oc->bhwo", xs, p1.w.double()[:, :Cin]) * p1.scale.double() + p1.shift.double() + rs)
    zd = torch.relu(torch.einsum("bhwc,oc->bhwo", yd, p2.w.double()[:, :Cmid]) * p2.scale.double() + p2.shift.double())
    assert float((z1[:1, :8].double() - zd).abs().max() / zd.abs().max()) < 2e-6
    # what is not such a pair is refused, not approximated
    assert ops.conv2d_b2b(x, p1, None, p2) is None and ops.conv2d_b2b(x[:1, :4, :4].contiguous(), p1, res[:1, :4, :4].contiguous(), p2) is None


def test_the_squeeze_behind_a_fused_boundary_keeps_its_bits_at_every_batch_size(hip_model, oracle):
    """The second layer of a back-to-back pair runs bf16x3 in EVERY form (layers._Packable.b2b_second): a frame's res2 output is the same
    bits whether its batch is large enough for the one-launch form (>= 4 frames: one round of the chip) or not."""
    from articulation3d_amd import ops

    frames = torch.from_numpy(oracle.synthetic_frames(6, seed=77)).cuda()
    bu = hip_model.backbone.bottom_up
    assert [blk.conv1.b2b_second for blk in bu.res2] == [False, True, True] and not any(blk.conv1.b2b_second for blk in bu.res4)
    x4 = ops.preprocess_u8hwc(frames, hip_model.pixel_mean, hip_model.pixel_std)
    seen = []
    real = ops.conv2d_b2b
    ops.conv2d_b2b = lambda *a, **k: (seen.append(real(*a, **k)) or seen[-1])
    try:
        big = bu(x4)
        n_fused = sum(1 for p in seen if p is not None)
        seen.clear()
        small = [bu(x4[i:i + 2].contiguous()) for i in range(0, 6, 2)]
        n_small = sum(1 for p in seen if p is not None)
    finally:
        ops.conv2d_b2b = real
    assert n_fused == 2 and n_small == 0, (n_fused, n_small)  # (res2's two boundaries at 6 frames; none at 2)
    for name in ("res2", "res3", "res5"):
        assert torch.equal(big[name], torch.cat([s[name] for s in small])), name
    saved, ops.B2B_FUSED = ops.B2B_FUSED, False  # (the switch drops the bf16x3 pin with the fusion: the round-5 arithmetic, for A/B runs)
    try:
        old = bu(x4)
    finally:
        ops.B2B_FUSED = saved
    rel = float((old["res2"] - big["res2"]).abs().max() / big["res2"].abs().max())
    assert 0 < rel < 1e-5, rel


def test_plane_split_winograd_equals_the_one_launch_form_bit_for_bit(ops):
    """Small fp16x2 Winograd problems run one plane per workgroup into a3d_conv_desc.wino_m and fold afterwards (wino_fold_kernel): the
    same multiply-adds in the same order as the one-launch kernel -- outputs and recorded maxima are identical."""
    torch.manual_seed(11)
    cases = [(1, 30, 40, 256, 256), (2, 15, 20, 512, 512), (3, 9, 11, 256, 128), (1, 60, 80, 256, 256), (5, 14, 14, 256, 256), (1, 8, 10, 256, 256), (3, 37, 41, 256, 256)]
    for B, H, W, Cin, Cout in cases:
        x = torch.randn(B, H, W, Cin, device="cuda") * torch.logspace(-1, 1, B, device="cuda")[:, None, None, None]
        pk = ops.pack_conv(torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5), torch.randn(Cout) * 0.1, None, 1, 1, ops.ACT_RELU)
        a = ops.conv2d(x, pk, precision=3)
        assert ops.last_conv_variant().endswith("wino_fold_kernel"), ops.last_conv_variant()
        ops.WINO_PLANE_SPLIT = False
        try:
            b = ops.conv2d(x, pk, precision=3)
            assert ops.last_conv_variant().startswith("wino_gemm_h2w_kernel<") and "fold" not in ops.last_conv_variant()
        finally:
            ops.WINO_PLANE_SPLIT = True
        assert torch.equal(a, b), (B, H, W, Cin, Cout)
        assert torch.equal(ops.amax_of(a), ops.amax_of(b))
    big = torch.randn(64, 30, 40, 256, device="cuda")  # 150 blocks: the one-launch form stays
    pk = ops.pack_conv(torch.randn(256, 256, 3, 3) / 48, None, None, 1, 1, ops.ACT_NONE)
    ops.conv2d(big, pk, precision=3)
    assert "fold" not in ops.last_conv_variant()


@pytest.mark.parametrize("precision", [3, 2])
def test_row_major_and_direct_epilogues_store_the_same_bits(ops, precision):
    """The split-operand kernels pass their output tiles through LDS so that a store covers 8 rows x 128 B; tune 12 keeps the direct
    per-lane stores.  Same value per element either way -- residual, upsampled residual, ragged row / channel counts, several images
    per tile (per-image maxima), two filter taps widths."""
    torch.manual_seed(9)
    cases = [  # B, H, W, Cin, Cout, k, stride, residual kind
        (3, 23, 31, 64, 256, 1, 1, "same"), (2, 30, 40, 512, 256, 1, 1, "ups"), (37, 3, 3, 128, 96, 3, 1, None),
        (2, 24, 40, 64, 64, 3, 1, None), (2, 31, 17, 256, 512, 1, 2, None), (5, 9, 7, 32, 132, 1, 1, "same")]
    for B, H, W, Cin, Cout, k, st, rk in cases:
        x = torch.randn(B, H, W, Cin, device="cuda") * torch.logspace(-1, 1, B, device="cuda")[:, None, None, None]
        pk = ops.pack_conv(torch.randn(Cout, Cin, k, k) / (k * Cin ** 0.5), torch.randn(Cout) * 0.1, None, st, k // 2, ops.ACT_RELU)
        Ho, Wo = (H + 2 * (k // 2) - k) // st + 1, (W + 2 * (k // 2) - k) // st + 1
        kw = {}
        if rk == "same":
            kw["res"] = torch.randn(B, Ho, Wo, pk.cols, device="cuda")
        elif rk == "ups":
            kw["res"], kw["res_ups"] = torch.randn(B, Ho // 2, Wo // 2, pk.cols, device="cuda"), True
        b = ops.conv2d(x, pk, precision=precision, tune=12, **kw)
        assert ops.last_conv_variant().startswith(("conv_h2_kernel", "conv_x3_kernel")), ops.last_conv_variant()
        for tune in (10, 11):  # 128 x 64 and 128 x 128 tiles
            a = ops.conv2d(x, pk, precision=precision, tune=tune, **kw)
            assert ops.last_conv_variant().startswith(("conv_h2_kernel", "conv_x3_kernel")), ops.last_conv_variant()
            assert torch.equal(a, b), (B, H, W, Cin, Cout, k, st, rk, tune)
            if precision == 3:
                assert torch.equal(ops.amax_of(a), ops.amax_of(b))


def test_detector_results_do_not_depend_on_winograd_input_sharing(ops, hip_model, oracle):
    model = hip_model
    model.roi_heads.box_predictor.test_score_thresh = 0.3
    frames = torch.from_numpy(oracle.synthetic_frames(4, seed=77)).cuda()  # > streams.SMALL_BATCH frames: the sharing scope is on
    a = model.inference_batched(frames, want_masks=True)
    ops.WINO_SHARE_ENABLED = False
    try:
        b = model.inference_batched(frames, want_masks=True)
    finally:
        ops.WINO_SHARE_ENABLED = True
    for k in ("depth", "records", "rec_count", "masks", "planes", "boxes"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k


def test_conv_fused_epilogues(ops):
    torch.manual_seed(2)
    x = torch.randn(2, 64, 24, 40)
    w = torch.randn(256, 64, 1, 1) / 8
    bn = (torch.rand(256) + 0.5, torch.randn(256) * 0.1, torch.randn(256) * 0.1, torch.rand(256) + 0.5, 1e-5)
    r = torch.randn(2, 256, 24, 40)
    ref = F.relu(F.batch_norm(F.conv2d(x, w), bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5) + r)
    y = ops.conv2d(nhwc(x).cuda(), ops.pack_conv(w, None, bn, 1, 0, ops.ACT_RELU), res=nhwc(r).cuda())
    assert rel(y.permute(0, 3, 1, 2), ref) < 5e-6
    # FPN lateral: + nearest x2 upsampled coarser level
    x = torch.randn(2, 512, 30, 40)
    w = torch.randn(256, 512, 1, 1) / 22
    b = torch.randn(256)
    prev = torch.randn(2, 256, 15, 20)
    ref = F.conv2d(x, w, b) + F.interpolate(prev, scale_factor=2.0, mode="nearest")
    y = ops.conv2d(nhwc(x).cuda(), ops.pack_conv(w, b), res=nhwc(prev).cuda(), res_ups=True)
    assert rel(y.permute(0, 3, 1, 2), ref) < 5e-6
    # depth deconv: nearest x2 upsample of a channel concat, leaky / relu
    a, c2 = torch.randn(2, 128, 15, 20), torch.randn(2, 128, 15, 20)
    w = torch.randn(128, 256, 3, 3) / 48
    ref = F.leaky_relu(F.conv2d(F.interpolate(torch.cat([a, c2], 1), scale_factor=2, mode="nearest"), w, b[:128], padding=1), 0.01)
    y = ops.conv2d(nhwc(a).cuda(), ops.pack_conv(w, b[:128], None, 1, 1, ops.ACT_LEAKY), x2=nhwc(c2).cuda(), ups=True)
    assert rel(y.permute(0, 3, 1, 2), ref) < 5e-6
    # transposed conv 2x2 s2 as a pixel-shuffled GEMM
    x = torch.randn(5, 256, 14, 14)
    w = torch.randn(256, 256, 2, 2) / 16
    ref = F.relu(F.conv_transpose2d(x, w, b, stride=2))
    y = ops.conv2d(nhwc(x).cuda(), ops.pack_deconv2x2(w, b))
    assert rel(y.permute(0, 3, 1, 2), ref) < 5e-6


@pytest.mark.parametrize("Cin,Cout,res_mode", [(64, 256, "res"), (128, 192, "none"), (256, 64, "none"), (512, 256, "ups")])
def test_persistent_pointwise_kernel(ops, Cin, Cout, res_mode):
    """1x1 layers with many tiles run in the persistent kernel (conv_pw.hip): vs torch, and bit-identical to the
    one-tile-per-workgroup kernel (tune=5) -- same k order, same epilogue -- including ragged M / N tiles, in-place
    residual (y aliases res, the training step's gradient accumulation) and the ReLU gate."""
    torch.manual_seed(11)
    B, H, W = 7, 122, 158  # M = 134932 rows: not a multiple of 128
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, 1, 1) / Cin ** 0.5
    bn = (torch.rand(Cout) + 0.5, torch.randn(Cout) * 0.1, torch.randn(Cout) * 0.1, torch.rand(Cout) + 0.5, 1e-5)
    pk = ops.pack_conv(w, None, bn, 1, 0, ops.ACT_RELU)
    if Cout <= 64:  # 64-wide layers need more rows for two rounds of the persistent grid
        B = 14
        x = torch.randn(B, Cin, H, W)
    xd = nhwc(x).cuda()
    kw, ref = {}, F.batch_norm(F.conv2d(x, w), bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5)
    if res_mode == "res":
        r = torch.randn(B, Cout, H, W)
        kw, ref = dict(res=nhwc(r).cuda()), ref + r
    elif res_mode == "ups":
        x = x[:, :, :122, :158]
        r = torch.randn(B, Cout, H // 2, W // 2)
        kw, ref = dict(res=nhwc(r).cuda(), res_ups=True), ref + F.interpolate(r, scale_factor=2.0, mode="nearest")
    ref = F.relu(ref)
    y = ops.conv2d(xd, pk, precision=0, **kw)  # (explicit: the suite can be run with A3D_PRECISION=2, and this is fp32 kernel vs fp32 kernel)
    assert ops.last_conv_variant().startswith("conv_pw_kernel<"), ops.last_conv_variant()  # what the dispatcher launched
    y5 = ops.conv2d(xd, pk, tune=5, **kw)
    assert ops.last_conv_variant().startswith("conv_gemm_v2_kernel<"), ops.last_conv_variant()
    assert rel(y.permute(0, 3, 1, 2), ref) < 5e-6
    assert rel(ops.conv2d(xd, pk, **kw).permute(0, 3, 1, 2), ref) < 5e-6  # whatever arithmetic the session default selects
    assert torch.equal(y, y5)
    if res_mode == "res":  # in place: the output buffer is the residual
        buf = kw["res"].clone()
        gate = torch.randn_like(buf)
        yg = ops.conv2d(xd, pk, res=buf, out=buf, gate=gate, precision=0)
        assert yg.data_ptr() == buf.data_ptr() and torch.equal(buf, y * (gate > 0))


def test_opt_in_bf16_arithmetic_mode(ops, hip_model, oracle):
    """ops.DEFAULT_PRECISION = 1 (bench.py --precision bf16): plain conv / linear layers on the bf16 MFMA.  Not a parity
    mode -- the check is that it is wired through the whole detector and stays at autocast-level distance from fp32."""
    model = hip_model
    model.roi_heads.box_predictor.test_score_thresh = 0.5
    frames = torch.from_numpy(oracle.synthetic_frames(2, seed=3)).cuda()
    ref = model.inference_batched(frames)
    saved = ops.DEFAULT_PRECISION
    ops.DEFAULT_PRECISION = 1
    try:
        out = model.inference_batched(frames)
    finally:
        ops.DEFAULT_PRECISION = saved
    l2 = ((out.depth - ref.depth).norm() / ref.depth.norm()).item()
    print("bf16 mode: depth relative L2 distance from fp32 %.4f, detections %s vs %s" % (l2, out.rec_count.tolist(), ref.rec_count.tolist()))
    # random-init weights make the depth map a near-cancelling 576-term sum: bf16 rounding moves it by ~0.6 relative L2 here (measured);
    # the assertion only pins that the mode is different from fp32, finite and of the same scale
    assert 1e-4 < l2 < 1.0 and bool(torch.isfinite(out.depth).all())
    assert (out.rec_count - ref.rec_count).abs().max().item() <= 8
    again = model.inference_batched(frames)  # the switch is off again: fp32 results are reproduced bit for bit
    assert torch.equal(again.depth, ref.depth) and torch.equal(again.records, ref.records)


@pytest.mark.parametrize("case", ["1x1-res", "3x3-s2", "linear-96", "narrow"])
def test_bf16x3_kernel_is_fp32_grade(ops, case):
    """a3d_conv_desc.precision == 2 (csrc/conv_bf16x3.hip): fp32 operands split exactly into three bf16 terms, six bf16 MFMAs
    per k step.  Against a float64 evaluation its error must be no larger than the native fp32-MFMA kernel's (measured: ~15 %
    smaller), on ragged M / N tiles, padding taps, a residual epilogue, a linear layer and a 64-wide layer."""
    torch.manual_seed(21)
    B, H, W, Cin, Cout, k, s, res = {"1x1-res": (3, 37, 41, 256, 200, 1, 1, True), "3x3-s2": (2, 45, 51, 128, 136, 3, 2, False),
                                     "linear-96": (777, 1, 1, 96, 1024, 1, 1, False), "narrow": (2, 40, 30, 64, 64, 1, 1, False)}[case]
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5
    bias = torch.randn(Cout)
    pk = ops.pack_conv(w, bias, None, s, k // 2, ops.ACT_NONE)
    ref = F.conv2d(x.double(), w.double(), bias.double(), s, k // 2)
    kw = {}
    if res:
        r = torch.randn(ref.shape)
        kw, ref = dict(res=nhwc(r).cuda()), ref + r.double()
    xd = nhwc(x).cuda()
    l2 = lambda y: ((y.permute(0, 3, 1, 2).double().cpu() - ref).norm() / ref.norm()).item()
    e32 = l2(ops.conv2d(xd, pk, precision=0, wino=False, **kw))
    ex3 = l2(ops.conv2d(xd, pk, precision=2, **kw))
    print(f"{case}: relative L2 error vs float64: fp32 MFMA {e32:.3e}, bf16x3 {ex3:.3e}")
    assert ex3 < 1e-6 and ex3 <= 1.05 * e32


@pytest.mark.parametrize("shape", [(2, 61, 79, 64, 64), (2, 31, 41, 128, 200), (5, 14, 14, 256, 256)])
def test_winograd_split_operand_gemm_is_fp32_grade(ops, shape):
    """precision == 2 on a Winograd layer: F(2x2,3x3) whose 16-plane GEMM multiplies exact 3-way bf16 splits of V and U
    (csrc/conv_wino.hip, 2x).  Against float64 its error is no larger than the fp32 Winograd form's (odd sizes, ragged channel
    tiles, the per-ROI 14x14 maps)."""
    torch.manual_seed(23)
    B, H, W, Cin, Cout = shape
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, 3, 3) / (9 * Cin) ** 0.5
    bias = torch.randn(Cout)
    pk = ops.pack_conv(w, bias, None, 1, 1, ops.ACT_RELU)
    ref = F.relu(F.conv2d(x.double(), w.double(), bias.double(), 1, 1))
    xd = nhwc(x).cuda()
    l2 = lambda y: ((y.permute(0, 3, 1, 2).double().cpu() - ref).norm() / ref.norm()).item()
    e32, ex3 = l2(ops.conv2d(xd, pk, precision=0)), l2(ops.conv2d(xd, pk, precision=2))
    assert pk.w_wino_x3 is not None  # it did take the Winograd route
    print(f"{shape}: relative L2 error vs float64: fp32 Winograd {e32:.3e}, split-operand Winograd {ex3:.3e}")
    assert ex3 < 1e-6 and ex3 <= 1.05 * e32


def test_bf16x3_mode_through_the_detector(ops, hip_model, oracle):
    """ops.DEFAULT_PRECISION = 2 (A3D_PRECISION=2 / bench.py --precision bf16x3) routes the non-Winograd conv / linear layers
    through the bf16x3 kernel.  It is an fp32-grade mode: features, depth and head outputs stay within the fp32 path's own
    parity tolerance of the fp32 run, the detections are the same set.  (The whole parity suite also passes with
    A3D_PRECISION=2 exported; the mode stays opt-in so that the headline is plain fp32-MFMA arithmetic.)"""
    model = hip_model
    model.roi_heads.box_predictor.test_score_thresh = 0.5
    frames = torch.from_numpy(oracle.synthetic_frames(2, seed=3)).cuda()
    saved = ops.DEFAULT_PRECISION
    try:
        ops.DEFAULT_PRECISION = 0
        ref = model.inference_batched(frames)
        ops.DEFAULT_PRECISION = 2
        out = model.inference_batched(frames)
    finally:
        ops.DEFAULT_PRECISION = saved
    l2 = ((out.depth - ref.depth).norm() / ref.depth.norm()).item()
    print("bf16x3 mode: depth relative L2 distance from fp32 %.2e, detections %s vs %s" % (l2, out.rec_count.tolist(), ref.rec_count.tolist()))
    # two fp32-grade evaluations of the random-init depth decoder (a near-cancelling sum, see the bf16 test above) sit ~1e-4
    # apart -- the distance the fp32 HIP path has from the fp32 oracle; plain bf16 arithmetic moves the same map by 0.6
    assert l2 < 5e-4
    assert torch.equal(out.rec_count, ref.rec_count)


@pytest.mark.parametrize("hw", [(480, 640), (96, 128), (61, 75), (250, 330), (14, 18)])
def test_fused_stem_is_the_two_launches_bit_for_bit(ops, hw):
    """Round 4: a3d_stem_conv_pool (7x7 s2 conv + BN + ReLU + 3x3 s2 max-pool, input patch resident in LDS, filter stationary in registers)
    against the stem conv launch followed by the pool launch: the same bits -- chunk order, product terms, un-scaling, epilogue and the
    pool's comparison order are shared -- and the same recorded maxima; odd sizes, a map smaller than one tile, a NaN pixel."""
    if ops.DEFAULT_PRECISION != 3:
        pytest.skip("the fused stem belongs to the fp16x2 arithmetic")
    torch.manual_seed(hw[0])
    B = 3 if hw[0] < 400 else 2
    x = torch.rand(B, 3, *hw) * 255 - 110
    x[1] *= 3.0  # (per-image scales differ)
    w = torch.randn(64, 3, 7, 7) / 12
    bn = (torch.rand(64) + 0.5, torch.randn(64) * 0.1, torch.randn(64) * 0.1, torch.rand(64) + 0.5, 1e-5)
    pk = ops.pack_stem(w, bn)
    pk.act = ops.ACT_RELU
    x4 = ops.preprocess_f32chw(x.cuda(), (0, 0, 0), (1, 1, 1))
    two = ops.maxpool3x3s2(ops.conv2d(x4, pk))
    assert ops.last_conv_variant() == "conv_h2_kernel<1> stem", ops.last_conv_variant()
    one = ops.stem_pool(x4, pk)
    assert one is not None and ops.last_conv_variant() == "stem_pool_kernel", ops.last_conv_variant()
    assert one.shape == two.shape and torch.equal(one, two)
    assert torch.equal(ops.amax_of(one), ops.amax_of(two))
    ref64 = F.max_pool2d(F.relu(F.batch_norm(F.conv2d(x.double(), w.double(), None, 2, 3), bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(),
                                             False, 0.0, 1e-5)), 3, 2, 1)
    assert rel(one.permute(0, 3, 1, 2).double(), ref64) < 2e-6
    xn = x4.clone()
    xn[0, hw[0] // 2, hw[1] // 2, 1] = float("nan")  # a NaN input pixel: every conv output it reaches is NaN, and a NaN wins its pool window
    a, b = ops.maxpool3x3s2(ops.conv2d(xn, pk)), ops.stem_pool(xn, pk)
    # (NaNs compare by position: the fused pool clears a NaN's sign bit, the only bits that may differ)
    assert bool(torch.isnan(b).any()) and torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))


def test_stem_pool_resize_small_ops(ops):
    torch.manual_seed(3)
    x = torch.rand(2, 3, 96, 128) * 255 - 110
    w = torch.randn(64, 3, 7, 7) / 12
    bn = (torch.rand(64) + 0.5, torch.randn(64) * 0.1, torch.randn(64) * 0.1, torch.rand(64) + 0.5, 1e-5)
    ref = F.relu(F.batch_norm(F.conv2d(x, w, None, 2, 3), bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-5))
    y = ops.conv2d(ops.preprocess_f32chw(x.cuda(), (0, 0, 0), (1, 1, 1)), ops.pack_stem(w, bn))
    assert rel(y.permute(0, 3, 1, 2), ref) < 5e-6
    # the stem in each arithmetic (its own loader in both GEMM kernels), odd image sizes included: against float64
    for hw in ((96, 128), (61, 75)):
        xs = torch.rand(2, 3, *hw) * 255 - 110
        ref64 = F.relu(F.batch_norm(F.conv2d(xs.double(), w.double(), None, 2, 3), bn[2].double(), bn[3].double(), bn[0].double(), bn[1].double(), False, 0.0, 1e-5))
        x4 = ops.preprocess_f32chw(xs.cuda(), (0, 0, 0), (1, 1, 1))
        for prec, name in ((0, "conv_gemm_v2_kernel"), (2, "conv_x3_kernel<1> stem")):
            ys = ops.conv2d(x4, ops.pack_stem(w, bn), precision=prec)
            assert ops.last_conv_variant().startswith(name), ops.last_conv_variant()
            assert rel(ys.permute(0, 3, 1, 2).double(), ref64) < 2e-6, (hw, prec)
    assert rel(ops.maxpool3x3s2(y).permute(0, 3, 1, 2), F.max_pool2d(ref, 3, 2, 1)) < 5e-6
    assert torch.equal(ops.subsample2(y).cpu(), y.cpu()[:, ::2, ::2])
    u8 = torch.randint(0, 256, (2, 32, 64, 3), dtype=torch.uint8)
    mean, std = (103.53, 116.28, 123.675), (1.0, 1.0, 1.0)
    got = ops.preprocess_u8hwc(u8.cuda(), mean, std).cpu()
    assert torch.equal(got[..., :3], u8.float() - torch.tensor(mean)) and float(got[..., 3].abs().sum()) == 0
    x = torch.randn(2, 128, 16, 20)
    assert rel(ops.resize_bilinear(nhwc(x).cuda(), 15, 20).permute(0, 3, 1, 2), F.interpolate(x, size=(15, 20), mode="bilinear", align_corners=False)) < 1e-6
    x = torch.randn(2, 1, 24, 32)
    assert rel(ops.resize_bilinear(nhwc(x).cuda(), 48, 64).permute(0, 3, 1, 2), F.interpolate(x, size=(48, 64), mode="bilinear", align_corners=False)) < 1e-6
    x = torch.randn(2, 64, 24, 32)
    w = torch.randn(1, 64, 3, 3) / 24
    y = ops.conv3x3_to1(nhwc(x).cuda(), w[0].permute(1, 2, 0).contiguous().cuda(), 0.3)
    assert rel(y[:, None], F.conv2d(x, w, torch.tensor([0.3]), padding=1)) < 5e-6


@pytest.mark.parametrize("shape", [(2, 30, 40, 128, 256), (1, 15, 21, 64, 36), (3, 14, 14, 256, 256), (1, 2, 2, 32, 64)])
def test_winograd_f2x2_3x3_vs_torch(ops, shape):
    """3x3 s1 p1 layers run as Winograd F(2x2,3x3) (odd sizes = partial tiles, ragged channels)."""
    B, H, W, Cin, Cout = shape
    torch.manual_seed(8)
    x = torch.randn(B, Cin, H, W)
    w = torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    bn = (torch.rand(Cout) + 0.5, torch.randn(Cout) * 0.1, torch.randn(Cout) * 0.1, torch.rand(Cout) + 0.5, 1e-3)
    b = torch.randn(Cout)
    ref = F.leaky_relu(F.batch_norm(F.conv2d(x, w, b, padding=1), bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-3), 0.01)
    p = ops.pack_conv(w, b, bn, 1, 1, ops.ACT_LEAKY)
    assert p.w_wino is not None and p.w_wino.shape == (16, p.cols, Cin)
    yw = ops.conv2d(nhwc(x).cuda(), p, wino=True)
    yd = ops.conv2d(nhwc(x).cuda(), p, wino=False)
    assert rel(yw[..., :Cout].permute(0, 3, 1, 2), ref) < 1e-5 and rel(yd[..., :Cout].permute(0, 3, 1, 2), ref) < 5e-6
    ref64 = F.leaky_relu(F.batch_norm(F.conv2d(x.double(), w.double(), b.double(), padding=1), bn[2].double(), bn[3].double(),
                                      bn[0].double(), bn[1].double(), False, 0.0, 1e-3), 0.01)
    # fp32 error budget against float64: Winograd stays within a small factor of the direct form
    assert rel(yw[..., :Cout].permute(0, 3, 1, 2).double(), ref64) < 6 * rel(yd[..., :Cout].permute(0, 3, 1, 2).double(), ref64) + 1e-6


def test_upsampled_conv_as_four_source_grid_phases(ops):
    """conv3x3(pad 1) over nearest-x2 upsample(cat(a, b)) == four 2x2 convs with pre-summed taps (depth decoder)."""
    torch.manual_seed(9)
    a, c2 = torch.randn(2, 128, 15, 20), torch.randn(2, 128, 15, 20)
    w = torch.randn(128, 256, 3, 3) / 48
    b = torch.randn(128)
    bn = (torch.rand(128) + 0.5, torch.randn(128) * 0.1, torch.randn(128) * 0.1, torch.rand(128) + 0.5, 1e-3)
    ref = F.relu(F.batch_norm(F.conv2d(F.interpolate(torch.cat([a, c2], 1), scale_factor=2, mode="nearest"), w, b, padding=1),
                              bn[2], bn[3], bn[0], bn[1], False, 0.0, 1e-3))
    phases = ops.pack_conv_ups_phases(w, b, bn, ops.ACT_RELU)
    assert [p.phase for p in phases] == [1, 2, 3, 4] and phases[0].Kpad == 4 * 256
    y = ops.conv2d_ups(nhwc(a).cuda(), phases, x2=nhwc(c2).cuda())
    assert y.shape == (2, 30, 40, 128) and rel(y.permute(0, 3, 1, 2), ref) < 5e-6
    # single source, odd spatial size
    a = torch.randn(1, 64, 7, 9)
    w = torch.randn(64, 64, 3, 3) / 24
    ref = F.conv2d(F.interpolate(a, scale_factor=2, mode="nearest"), w, None, padding=1)
    y = ops.conv2d_ups(nhwc(a).cuda(), ops.pack_conv_ups_phases(w))
    assert rel(y.permute(0, 3, 1, 2), ref) < 5e-6


@pytest.mark.parametrize("case", [(3, 30, 40, 128, 128, 128), (2, 61, 79, 256, 0, 64), (5, 15, 20, 256, 256, 128), (1, 8, 10, 128, 128, 32)],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_fused_upsampled_conv_equals_the_four_phase_launches(ops, case):
    """The depth decoder's upsampled convs as ONE launch over the 9 distinct taps (a3d_conv_desc.phase == 5: every wave holds all
    four phases and skips the tap-phase blocks that are zero): outputs and recorded maxima equal the four-launch form bit for bit,
    and both are fp32-grade against the float64 convolution of the upsampled tensor."""
    B, H, W, C1, C2, Cout = case
    torch.manual_seed(13)
    a = torch.randn(B, H, W, C1, device="cuda") * torch.logspace(-1, 1, B, device="cuda")[:, None, None, None]
    c2 = torch.randn(B, H, W, C2, device="cuda") if C2 else None
    w = torch.randn(Cout, C1 + C2, 3, 3) / (3 * (C1 + C2) ** 0.5)
    b = torch.randn(Cout) * 0.1
    bn = (torch.rand(Cout) + 0.5, torch.randn(Cout) * 0.1, torch.randn(Cout) * 0.1, torch.rand(Cout) + 0.5, 1e-3)
    phases = ops.pack_conv_ups_phases(w, b, bn, ops.ACT_LEAKY)
    four = ops.conv2d_ups(a, phases, x2=c2, fused=False)
    assert ops.last_conv_variant().startswith("conv_h2_kernel")
    one = ops.conv2d_ups(a, phases, x2=c2, fused=True, tune=15)  # (the tap-outer kernel: the form that claims the four launches' bits)
    assert ops.last_conv_variant() == "conv_h2w_kernel ph4", ops.last_conv_variant()
    assert one.shape == (B, 2 * H, 2 * W, Cout)
    assert torch.equal(one, four) and torch.equal(one._a3d_amax, four._a3d_amax)
    dflt = ops.conv2d_ups(a, phases, x2=c2)  # the default since round 4: the patch-resident kernel (another reduction order: fp32 rounding)
    assert ops.last_conv_variant().startswith("conv_ph4p_kernel<"), ops.last_conv_variant()
    for i in range(B):
        assert float((dflt[i] - one[i]).abs().max() / one[i].abs().max()) < 2e-6
    xin = a if c2 is None else torch.cat([a, c2], 3)
    up = F.interpolate(xin.permute(0, 3, 1, 2).double(), scale_factor=2, mode="nearest")
    ref = F.batch_norm(F.conv2d(up, w.double().cuda(), b.double().cuda(), padding=1), bn[2].double().cuda(), bn[3].double().cuda(),
                       bn[0].double().cuda(), bn[1].double().cuda(), False, 0.0, 1e-3)
    ref = F.leaky_relu(ref, 0.01).permute(0, 2, 3, 1)
    for i in range(B):
        assert float((one[i].double() - ref[i]).norm() / ref[i].norm()) < 2e-6


@pytest.mark.parametrize("case", [(3, 30, 40, 128, 128, 128), (2, 61, 79, 256, 0, 64), (2, 15, 20, 256, 256, 128), (2, 120, 160, 128, 128, 64)],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_fused_upsampled_conv_in_the_bf16x3_arithmetic(ops, case):
    """Round 5: the fused four-phase launch instantiated for the like-for-like arithmetic (exact 3-way bf16 splits, six products per step:
    conv_x3w_kernel<false, true>) -- the depth decoder's deconv layers (depth_head.py:40-46,58-68) when the package runs bf16x3, and
    every layer the precision audit pins.  Same term order per accumulator as the four phase launches of conv_x3_kernel: equal bits;
    fp32-grade against the float64 convolution of the upsampled tensor; a frame's result does not depend on its batch."""
    B, H, W, C1, C2, Cout = case
    torch.manual_seed(17)
    a = torch.randn(B, H, W, C1, device="cuda") * torch.logspace(-1, 1, B, device="cuda")[:, None, None, None]
    c2 = torch.randn(B, H, W, C2, device="cuda") if C2 else None
    w = torch.randn(Cout, C1 + C2, 3, 3) / (3 * (C1 + C2) ** 0.5)
    b = torch.randn(Cout) * 0.1
    phases = ops.pack_conv_ups_phases(w, b, None, ops.ACT_RELU)
    with arithmetic(ops, 2):
        four = ops.conv2d_ups(a, phases, x2=c2, fused=False)
        assert ops.last_conv_variant().startswith("conv_x3_kernel"), ops.last_conv_variant()
        one = ops.conv2d_ups(a, phases, x2=c2)
        assert ops.last_conv_variant() == "conv_x3w_kernel ph4", ops.last_conv_variant()
        alone = ops.conv2d_ups(a[-1:].contiguous(), phases, x2=None if c2 is None else c2[-1:].contiguous())
    assert one.shape == (B, 2 * H, 2 * W, Cout) and torch.equal(one, four) and torch.equal(alone[0], one[-1])
    xin = a if c2 is None else torch.cat([a, c2], 3)
    up = F.interpolate(xin[-1:].permute(0, 3, 1, 2).double(), scale_factor=2, mode="nearest")
    ref = F.relu(F.conv2d(up, w.double().cuda(), b.double().cuda(), padding=1)).permute(0, 2, 3, 1)
    assert float((one[-1].double() - ref[0]).norm() / ref[0].norm()) < 1e-6


@pytest.mark.parametrize("splitk", [1, 7, 64])
def test_linear_splitk_chw_reorder(ops, splitk):
    torch.manual_seed(4)
    x = torch.randn(7, 256, 14, 14)
    w = torch.randn(1024, 256 * 14 * 14) / 224
    b = torch.randn(1024)
    ref = F.relu(F.linear(x.flatten(1), w, b))
    p = ops.pack_linear(w, b, chw=(256, 14, 14), act=ops.ACT_RELU)
    y = ops.linear(nhwc(x).reshape(7, -1).cuda(), p, splitk=splitk)
    assert rel(y, ref) < 1e-5
    if splitk > 1:  # slabs are summed in slice order: bitwise reproducible
        assert torch.equal(y, ops.linear(nhwc(x).reshape(7, -1).cuda(), p, splitk=splitk))


def test_ragged_row_count_from_device(ops):
    """m_dev: tiles past the live row count exit at once and leave the tail untouched."""
    torch.manual_seed(5)
    x = torch.randn(300, 64)
    w = torch.randn(128, 64) / 8
    p = ops.pack_linear(w, None)
    live = torch.tensor([130], dtype=torch.int32).cuda()
    out = torch.full((300, 1, 1, 128), -7.0).cuda()
    ops.conv2d(x.view(300, 1, 1, 64).cuda(), p, m_dev=live, out=out)
    y = out.view(300, 128).cpu()
    assert rel(y[:130], F.linear(x[:130], w)) < 5e-6
    assert bool((y[256:] == -7.0).all())  # whole tiles past the count were never written


# ------------------------------------------------------------------------------------------ ROIAlign / NMS units
@pytest.mark.parametrize("P,ratio,aligned", [(7, 0, True), (14, 2, False), (14, 0, False)])
def test_roi_align_fpn_vs_oracle(ops, oracle, P, ratio, aligned):
    torch.manual_seed(6)
    B = 2
    feats = {n: torch.randn(B, 256, 480 // s, 640 // s) for n, s in zip(NAMES[:4], (4, 8, 16, 32))}
    rng = np.random.default_rng(P)
    box_lists = []
    for b in range(B):
        n = 40 + 7 * b
        side = np.exp(rng.uniform(np.log(4), np.log(600), n))  # covers all four pyramid levels
        ar = np.exp(rng.uniform(-1, 1, n))
        w_, h_ = side * np.sqrt(ar), side / np.sqrt(ar)
        cx, cy = rng.uniform(0, 640, n), rng.uniform(0, 480, n)
        bx = np.stack([cx - w_ / 2, cy - h_ / 2, cx + w_ / 2, cy + h_ / 2], 1)
        bx[0] = [10, 10, 10, 10]  # zero-area box
        bx[1] = [-50, -40, 700, 500]  # larger than the image
        box_lists.append(torch.tensor(bx, dtype=torch.float32))
    ref = oracle.roi_pool_fpn(feats, box_lists, P, ratio, aligned)
    R = max(len(b) for b in box_lists)
    boxes = torch.zeros(B, R, 4)
    for i, bl in enumerate(box_lists):
        boxes[i, : len(bl)] = bl
    count = torch.tensor([len(b) for b in box_lists], dtype=torch.int32).cuda()
    off = ops.count_offsets(count, R)
    total = sum(len(b) for b in box_lists)
    out, lvl = ops.roi_align_fpn([nhwc(feats[n]).cuda() for n in NAMES[:4]], [0.25, 0.125, 0.0625, 0.03125], boxes.cuda(), count,
                                 P, ratio, aligned, row_offset=off, rows=total, want_level=True)
    assert torch.equal(lvl.cpu().long(), oracle.assign_levels(torch.cat(box_lists)))  # level indices bit-exact
    assert set(lvl.cpu().tolist()) == {0, 1, 2, 3}
    assert rel(out.permute(0, 3, 1, 2), ref) < 1e-5


def _nms_groups(seed, G_, n_max, dup=True):
    rng = np.random.default_rng(seed)
    boxes = torch.zeros(G_, 1024, 4)
    valid = torch.zeros(G_, 1024, dtype=torch.int32)
    ns = torch.zeros(G_, dtype=torch.int32)
    for g in range(G_):
        n = int(rng.integers(0, n_max + 1)) if g else n_max
        cx, cy = rng.uniform(0, 640, n), rng.uniform(0, 480, n)
        w_, h_ = rng.uniform(8, 200, n), rng.uniform(8, 200, n)
        b = np.stack([cx - w_ / 2, cy - h_ / 2, cx + w_ / 2, cy + h_ / 2], 1).astype(np.float32)
        if dup and n > 10:
            b[5] = b[2]  # exact duplicate
            b[7, :] = [100, 100, 100, 140]  # zero-area
        boxes[g, :n] = torch.from_numpy(b)
        v = rng.random(n) > 0.05
        valid[g, :n] = torch.from_numpy(v.astype(np.int32))
        ns[g] = n
    return boxes, valid, ns


@pytest.mark.parametrize("thr", [0.5, 0.7])
def test_group_nms_keep_masks_bit_exact(ops, oracle, thr):
    boxes, valid, ns = _nms_groups(11, 12, 1000)
    keep = ops.group_nms(boxes.cuda(), valid.cuda(), ns.cuda(), thr).cpu()
    for g in range(boxes.shape[0]):
        n = int(ns[g])
        v = valid[g, :n].bool()
        exp = torch.zeros(n, dtype=torch.bool)
        exp[v] = oracle.nms_sorted(boxes[g, :n][v], torch.zeros(int(v.sum()), dtype=torch.int64), thr)
        assert torch.equal(keep[g, :n].bool(), exp), f"group {g}"
        assert int(keep[g, n:].sum()) == 0
    # idempotence at full size: NMS of the kept boxes keeps all of them
    g = 0
    kept = boxes[g, : int(ns[g])][keep[g, : int(ns[g])].bool()]
    b2 = torch.zeros(1, 1024, 4)
    b2[0, : len(kept)] = kept
    k2 = ops.group_nms(b2.cuda(), torch.ones(1, 1024, dtype=torch.int32).cuda(), torch.tensor([len(kept)], dtype=torch.int32).cuda(), thr).cpu()
    assert int(k2[0, : len(kept)].sum()) == len(kept)


def test_group_nms_edge_cases(ops):
    boxes = torch.zeros(3, 1024, 4)
    boxes[1, :3] = torch.tensor([[0.0, 0, 10, 10], [0, 0, 10, 10], [20, 20, 30, 30]])
    boxes[2, :2] = torch.tensor([[0.0, 0, 10, 10], [1, 1, 11, 11]])
    valid = torch.ones(3, 1024, dtype=torch.int32)
    valid[2, 0] = 0  # an invalid box neither survives nor suppresses
    ns = torch.tensor([0, 3, 2], dtype=torch.int32)
    keep = ops.group_nms(boxes.cuda(), valid.cuda(), ns.cuda(), 0.5).cpu()
    assert int(keep[0].sum()) == 0
    assert keep[1, :3].tolist() == [1, 0, 1]
    assert keep[2, :2].tolist() == [0, 1]


# ------------------------------------------------------------------------------------------ golden vectors (reference modules)
def test_golden_paste_masks_through_hip(ops, golden_dir):
    from articulation3d_amd.modeling.postprocessing import paste_masks_in_image

    g = np.load(os.path.join(golden_dir, "paste_masks.npz"))
    masks, boxes, hw = G.paste_case()
    # the kernel vectorises 16 pixels: the 120x160 fixture image satisfies W % 16 == 0
    out = paste_masks_in_image(masks.cuda(), boxes.cuda(), hw, threshold=0.5).cpu().numpy()
    ref = np.unpackbits(g["packed"])[: int(np.prod(g["shape"]))].reshape(g["shape"]).astype(bool)
    assert (out != ref).sum() == 0  # bit-exact against the REFERENCE's paste_masks_in_image


def _close_to_truth(got, g, key, tol=1e-4):
    """`got` vs the REFERENCE module evaluated in float64 (the fixture's *_f64 arrays).  North-star tolerance:
    1e-4 relative to the magnitude of the quantity (unit vectors / O(1) offsets); and the HIP fp32 result may not
    be more than a small factor further from the truth than the reference's own fp32 CPU result is."""
    truth, ref32 = g[key + "_f64"], g[key]
    scale = max(1.0, float(np.abs(truth).max()))
    err = float(np.abs(got - truth).max())
    ref_err = float(np.abs(ref32 - truth).max())
    assert err < tol * scale, (key, err)
    assert err < 8 * ref_err + 2e-6, (key, err, ref_err)


def test_golden_plane_axis_heads_through_hip(ops, hip_model, golden_dir):
    import copy

    rh = hip_model.roi_heads
    x = G.head_input()
    xr = nhwc(x).cuda()
    ph = copy.deepcopy(rh.plane_head)
    ph.load_state_dict({k.replace("roi_heads.plane_head.", ""): v for k, v in G.head_params("plane").items()})
    g = np.load(os.path.join(golden_dir, "plane_head.npz"))
    _close_to_truth(ph.cuda().forward_rows(xr).cpu().numpy(), g, "pred_plane")
    ah = copy.deepcopy(rh.axis_head)
    ah.load_state_dict({k.replace("roi_heads.axis_head.", ""): v for k, v in G.head_params("axis").items()})
    g = np.load(os.path.join(golden_dir, "axis_head.npz"))
    rot, tran = ah.cuda().forward_rows(xr)
    _close_to_truth(rot.cpu().numpy(), g, "pred_rot_axis")
    _close_to_truth(tran.cpu().numpy(), g, "pred_tran_axis")


def test_golden_depth_head_through_hip(ops, hip_model, golden_dir):
    from articulation3d_amd.modeling.depth_head import PlaneRCNNDepthHead
    from conftest import make_cfg

    dh = PlaneRCNNDepthHead(make_cfg(0.7)).cuda().eval()
    missing, unexpected = dh.load_state_dict({k.replace("depth_head.", ""): v for k, v in G.head_params("depth").items()}, strict=False)
    assert not unexpected
    g = np.load(os.path.join(golden_dir, "depth_head.npz"))
    feats = {k: nhwc(v).cuda() for k, v in G.depth_features().items()}
    d = dh.forward_nhwc(feats).cpu()
    assert tuple(d.shape) == tuple(g["shape"])
    _close_to_truth(d[:, ::16, ::16].numpy(), g, "depth_strided")
    assert abs(float(d.double().sum()) - float(g["depth_sum"])) < 1e-4 * float(g["depth_abs_sum"])


# ------------------------------------------------------------------------------------------ the whole path, stage-wise
def _merge_expected(g, g0, ng, K):
    items = []
    for gl in range(ng):
        n = int(g["n"][g0 + gl])
        keep = g["keep"][g0 + gl, :n].bool()
        for r in keep.nonzero().squeeze(1).tolist():
            items.append((-float(g["scores"][g0 + gl, r]), int(g["pos"][g0 + gl, r]), gl, r))
    items.sort(key=lambda t: (t[0], t[1]))
    items = items[:K]
    boxes = torch.stack([g["boxes"][g0 + gl, r] for _, _, gl, r in items]) if items else torch.zeros(0, 4)
    return boxes, [it[2] for it in items]


MODES = {3: "fp16x2", 2: "bf16x3", 0: "fp32"}
_ORACLE_STAGE = {}


class arithmetic:
    """`with arithmetic(ops, p):` -- the library's module-default arithmetic (ops.DEFAULT_PRECISION) for the duration of the block.
    The stage tests run once per fp32-grade mode: 3 = fp16x2 (the default), 2 = bf16x3, 0 = fp32-input MFMA."""

    def __init__(self, ops, p):
        self.ops, self.p = ops, p

    def __enter__(self):
        self.saved, self.ops.DEFAULT_PRECISION = self.ops.DEFAULT_PRECISION, self.p

    def __exit__(self, *exc):
        self.ops.DEFAULT_PRECISION = self.saved
        return False


@pytest.fixture(scope="module", params=[3, 2, 0], ids=lambda p: MODES[p])
def staged(request, ops, hip_model, oracle, oracle_params):
    """Runs the HIP path on two synthetic frames in one arithmetic mode, keeping every intermediate.  (The mode is set only while
    the HIP launches of a fixture / test run -- `arithmetic` -- never for the lifetime of the fixture: tests that do not use the
    fixture may be scheduled between its users.)"""
    O, P, model = oracle, oracle_params, hip_model
    nf = 2
    frames = O.synthetic_frames(nf)
    fr = torch.from_numpy(frames).cuda()
    ocfg = O.OracleCfg(score_thresh=0.0)
    model.roi_heads.box_predictor.test_score_thresh = 0.0
    if "x" not in _ORACLE_STAGE:  # the oracle's side does not depend on the HIP arithmetic: once for the three modes
        x, _ = O.preprocess(O.frames_to_chw(frames), ocfg)
        _ORACLE_STAGE.update(x=x, ofeats=O.backbone(x, P))
    s = dict(nf=nf, frames=frames, fr=fr, ocfg=ocfg, ofeats=_ORACLE_STAGE["ofeats"], x=_ORACLE_STAGE["x"], precision=request.param)
    with arithmetic(ops, request.param):
        s["x4"] = ops.preprocess_u8hwc(fr, model.pixel_mean, model.pixel_std)
        s["feats"] = model.backbone.forward_nhwc(s["x4"])
        rpn = model.proposal_generator
        s["heads"] = rpn.rpn_head.forward_nhwc([s["feats"][f] for f in rpn.in_features])
        s["depth"] = model.depth_head.forward_nhwc(s["feats"])
        s["gfeats"] = {k: s["feats"][k].permute(0, 3, 1, 2).contiguous().cpu() for k in NAMES}
        s["props"] = rpn.forward_batched(s["feats"], HW, heads=s["heads"], return_groups=True)
        torch.cuda.synchronize()
    return s


def test_stage_backbone_rpnhead_depth_end_to_end(staged, oracle, oracle_params):
    s, O, P = staged, oracle, oracle_params
    assert rel(s["x4"][..., :3].permute(0, 3, 1, 2), s["x"]) == 0.0
    for k in NAMES:  # ~50 fp32 layers deep; both sides are fp32 with different summation orders
        assert rel(s["feats"][k].permute(0, 3, 1, 2), s["ofeats"][k]) < 2e-4, k
    ol, od = O.rpn_head(s["ofeats"], P)
    for l in range(5):
        assert rel(s["heads"][l][..., :3].reshape(s["nf"], -1), ol[l]) < 2e-4
        assert rel(s["heads"][l][..., 3:15].reshape(s["nf"], -1, 4), od[l]) < 2e-4
    assert rel(s["depth"], O.depth_head(s["ofeats"], P)) < 5e-4
    assert rel(s["depth"], O.depth_head(s["gfeats"], P)) < 2e-5  # same features in: the depth head itself


def test_stage_proposals_bit_exact_on_identical_heads(staged, oracle):
    s, O = staged, oracle
    nf, ocfg = s["nf"], s["ocfg"]
    heads = s["heads"]
    gl_ = [h[..., :3].reshape(nf, -1).cpu() for h in heads]
    gd_ = [h[..., 3:15].reshape(nf, -1, 4).cpu() for h in heads]
    feat_hw = [tuple(s["feats"][k].shape[1:3]) for k in NAMES]
    oprops, ogroups = O.rpn_select(gl_, gd_, feat_hw, [HW] * nf, ocfg, return_groups=True)
    pb, pl, plv, ppos, pc, g = s["props"]
    g = {k: v.cpu() for k, v in g.items()}
    for b in range(nf):
        for l in range(5):
            gi, og = b * 5 + l, ogroups[b][l]
            k = len(og["scores"])
            assert int(g["n"][gi]) == k
            assert torch.equal(g["scores"][gi, :k], og["scores"])  # top-k selection + order, bit-exact
            assert (g["boxes"][gi, :k] - og["boxes"]).abs().max() < 2e-3
            assert torch.equal(g["valid"][gi, :k].bool(), og["valid"])
            v = g["valid"][gi, :k].bool()
            okeep = torch.zeros(k, dtype=torch.bool)
            okeep[v] = O.nms_sorted(g["boxes"][gi, :k][v], torch.zeros(int(v.sum()), dtype=torch.int64), ocfg.rpn_nms_thresh)
            assert torch.equal(g["keep"][gi, :k].bool(), okeep)  # NMS keep mask, bit-exact
        exp_boxes, exp_lvl = _merge_expected(g, b * 5, 5, 1000)
        n = int(pc[b])
        assert n == len(exp_boxes) and torch.equal(pb[b, :n].cpu(), exp_boxes) and plv[b, :n].cpu().tolist() == exp_lvl
        assert float(pb[b, n:].abs().sum()) == 0
        sc = pl[b, :n].cpu()
        assert bool((sc[:-1] >= sc[1:]).all())  # score-descending
        ob = oprops[b][0]  # against the full CPU selection (differs only through exp ulps)
        assert n == len(ob) and (pb[b, :n].cpu() - ob).abs().max() < 2e-3


@pytest.fixture(scope="module")
def staged_box(staged, hip_model, oracle, oracle_params):
    s, O, P, model = staged, oracle, oracle_params, hip_model
    rh = model.roi_heads
    pb, _pl, _lv, _pos, pc, _g = s["props"]
    lv = [s["feats"][f] for f in rh.box_in_features]
    from articulation3d_amd import ops as _ops
    with arithmetic(_ops, s["precision"]):
        pooled = rh.box_pooler.forward_batched(lv, pb, pc)
        pred = rh.box_predictor(rh.box_head(pooled))
        rh.box_predictor.test_score_thresh = 0.0
        det = rh.box_predictor.inference_batched(pred, pb, pc, HW, return_groups=True)
        torch.cuda.synchronize()
    return dict(pooled=pooled, pred=pred, det=det, props_cpu=[pb[b, : int(pc[b])].cpu() for b in range(s["nf"])])


def test_stage_proposals_with_2000_candidates_per_level(ops, staged, oracle):
    """The training configuration's PRE_NMS_TOPK_TRAIN = 2000 (step1_bbox.yaml:21): 2048-slot selection groups, NMS with its
    suppression words in global memory, 16 384-key merge.  Same discrete results as the oracle on identical head outputs:
    count, level of every proposal, boxes to expf rounding, scores bit for bit."""
    s, O = staged, oracle
    nf = s["nf"]
    heads = s["heads"]
    rpn = None
    ocfg = O.OracleCfg(**{**s["ocfg"].__dict__, "rpn_pre_topk": 2000, "rpn_post_topk": 1000})
    gl_ = [h[..., :3].reshape(nf, -1).cpu() for h in heads]
    gd_ = [h[..., 3:15].reshape(nf, -1, 4).cpu() for h in heads]
    feat_hw = [tuple(s["feats"][k].shape[1:3]) for k in NAMES]
    oprops = O.rpn_select(gl_, gd_, feat_hw, [HW] * nf, ocfg)
    import math
    cell = torch.stack([O.cell_anchors(z, ocfg.anchor_ratios) for z in ocfg.anchor_sizes])
    pb, pl, plv, ppos, pc = ops.rpn_proposals(heads, [4, 8, 16, 32, 64], cell, HW, pre_topk=2000, post_topk=1000, nms_thresh=0.7, min_size=0.0,
                                              weights=(1.0, 1.0, 1.0, 1.0), scale_clamp=math.log(1000.0 / 16))
    for b in range(nf):
        ob, osc = oprops[b]
        n = int(pc[b])
        assert n == len(ob)
        assert torch.equal(pl[b, :n].cpu(), osc)
        assert (pb[b, :n].cpu() - ob).abs().max().item() < 1e-3
    # and the 1000-candidate setting still goes through the LDS-resident kernels with identical results
    pb1, pl1, _, _, pc1 = ops.rpn_proposals(heads, [4, 8, 16, 32, 64], cell, HW, pre_topk=1000, post_topk=1000, nms_thresh=0.7, min_size=0.0,
                                            weights=(1.0, 1.0, 1.0, 1.0), scale_clamp=math.log(1000.0 / 16))
    assert torch.equal(pb1, s["props"][0]) and torch.equal(pc1, s["props"][4])


def test_stage_box_head_and_detections(staged, staged_box, oracle, oracle_params):
    s, sb, O, P = staged, staged_box, oracle, oracle_params
    nf, ocfg = s["nf"], s["ocfg"]
    pb, _pl, _lv, _pos, pc, _g = s["props"]
    R = pb.shape[1]
    opooled = O.roi_pool_fpn(s["gfeats"], sb["props_cpu"], *ocfg.box_pool)
    gp = torch.cat([sb["pooled"][b * R: b * R + int(pc[b])] for b in range(nf)]).permute(0, 3, 1, 2)
    assert rel(gp, opooled) < 1e-5
    ocls, odl = O.box_predictor(O.box_head(opooled, P), P)
    gpred = torch.cat([sb["pred"][b * R: b * R + int(pc[b])] for b in range(nf)])
    assert rel(gpred[:, :3], ocls) < 1e-4 and rel(gpred[:, 3:11], odl) < 1e-4
    db, dsc, dcl, dpos, dcnt, g2 = sb["det"]
    g2 = {k: v.cpu() for k, v in g2.items()}
    for b in range(nf):
        n = int(pc[b])
        pr = sb["pred"][b * R: b * R + n].cpu()
        dec = O.apply_deltas(pr[:, 3:11], sb["props_cpu"][b], ocfg.box_weights, ocfg.scale_clamp)
        ob_, os_, oc_, _rows = O.fast_rcnn_inference_single(dec, F.softmax(pr[:, :3], dim=-1), HW, ocfg)
        exp_boxes, exp_cls = _merge_expected(g2, b * 2, 2, 100)
        nd = int(dcnt[b])
        assert nd == len(exp_boxes) == 100
        assert torch.equal(db[b, :nd].cpu(), exp_boxes) and dcl[b, :nd].cpu().tolist() == exp_cls
        assert nd == len(ob_) and (db[b, :nd].cpu() - ob_).abs().max() < 2e-3
        assert (dsc[b, :nd].cpu() - os_).abs().max() < 1e-6
        assert torch.equal(dcl[b, :nd].cpu().long(), oc_)  # class indices bit-exact
        for c in range(2):
            gi = b * 2 + c
            k = int(g2["n"][gi])
            v = g2["valid"][gi, :k].bool()
            okeep = torch.zeros(k, dtype=torch.bool)
            okeep[v] = O.nms_sorted(g2["boxes"][gi, :k][v], torch.zeros(int(v.sum()), dtype=torch.int64), ocfg.nms_thresh)
            assert torch.equal(g2["keep"][gi, :k].bool(), okeep)


def test_stage_roi_heads_paste_lsq_and_records(staged, staged_box, hip_model, oracle, oracle_params):
    from articulation3d_amd.modeling.roi_heads.roi_heads import BatchedDetections

    s, sb, O, P, model = staged, staged_box, oracle, oracle_params, hip_model
    nf, ocfg = s["nf"], s["ocfg"]
    db, dsc, dcl, _dpos, dcnt, _g2 = sb["det"]
    from articulation3d_amd import ops as _ops
    with arithmetic(_ops, s["precision"]):
        det = model.roi_heads.given_boxes_batched(s["feats"], BatchedDetections(db, dsc, dcl, dcnt, HW))
        torch.cuda.synchronize()
    dets_cpu = [db[b, : int(dcnt[b])].cpu() for b in range(nf)]
    assert det.total == sum(len(d) for d in dets_cpu) == 200
    om = O.mask_head(O.roi_pool_fpn(s["gfeats"], dets_cpu, *ocfg.mask_pool), P)
    assert rel(det.mask_prob[:, None], om) < 1e-4
    assert rel(det.pred_plane, O.plane_head(O.roi_pool_fpn(s["gfeats"], dets_cpu, *ocfg.plane_pool), P)) < 1e-4
    ora, ota = O.axis_head(O.roi_pool_fpn(s["gfeats"], dets_cpu, *ocfg.axis_pool), P)
    assert rel(det.pred_rot_axis, ora) < 1e-4 and rel(det.pred_tran_axis, ota) < 1e-4
    out = model._post_batched(det, s["depth"], HW, True, None)
    rays = O.k_inv_dot_xy1()
    rec = out.records.cpu()
    for b in range(nf):
        nd = int(dcnt[b])
        r0 = int(det.row_offset[b])
        d = dict(pred_boxes=db[b, :nd].cpu(), scores=dsc[b, :nd].cpu(), pred_classes=dcl[b, :nd].cpu().long(),
                 pred_masks=det.mask_prob[r0: r0 + nd, None].cpu(), pred_plane=det.pred_plane[r0: r0 + nd].cpu(), image_size=HW)
        o = O.detector_postprocess(d, HW[0], HW[1], ocfg)
        idx = out.keep[b, :nd].bool().cpu().nonzero().squeeze(1)
        assert len(idx) == len(o["scores"])
        assert (out.masks[b, idx].cpu().bool() != o["pred_masks"]).sum().item() == 0  # pasted masks bit-exact
        assert torch.equal(out.area[b, idx].cpu().long(), o["pred_masks"].sum((1, 2)))
        opo = O.override_depth(s["depth"][b].cpu(), o["pred_masks"], o["pred_plane"], rays)
        assert ((out.planes[b, idx].cpu() - opo).abs().max() / opo.abs().max()).item() < 1e-4  # north_star tolerance
        # packed records = the create_instances payload
        n = int(out.rec_count[b])
        assert n == len(idx)
        assert torch.equal(rec[b, :n, 0:4], out.boxes[b, idx].cpu()) and torch.equal(rec[b, :n, 4], dsc[b, idx].cpu())
        assert torch.equal(rec[b, :n, 5].long(), dcl[b, idx].cpu().long()) and torch.equal(rec[b, :n, 6:9], out.planes[b, idx].cpu())
        assert torch.equal(rec[b, :n, 9:12], det.pred_rot_axis[r0 + idx].cpu()) and torch.equal(rec[b, :n, 14:], det.mask_prob[r0 + idx].reshape(n, -1).cpu())
        assert float(rec[b, n:].abs().sum()) == 0


# ------------------------------------------------------------------------------------------ reference-signature path and edge cases
def test_reference_api_matches_batched_path(hip_model, oracle):
    from articulation3d_amd.utils.arti_vis import create_instances, PlaneRCNN_Branch

    model = hip_model
    model.roi_heads.box_predictor.test_score_thresh = 0.0
    frames = oracle.synthetic_frames(2, seed=7)
    out_b = model.inference_batched(torch.from_numpy(frames).cuda(), want_masks=True)
    inputs = [{"image": torch.as_tensor(f.transpose(2, 0, 1).astype("float32"))} for f in frames]
    outs_fast = model(inputs)  # list[{"instances", "depth"}] as planercnn.py:143-146 (uniform batch -> one batched pass)
    model.fast_reference_path = False
    try:
        outs = model(inputs)   # the module-by-module path (proposal generator / ROI heads / postprocess signatures)
    finally:
        model.fast_reference_path = True
    assert len(outs) == len(outs_fast) == 2
    for o, f in zip(outs, outs_fast):  # the two routes of the reference signature agree bit for bit
        a, c = o["instances"], f["instances"]
        assert torch.equal(a.pred_boxes.tensor, c.pred_boxes.tensor) and torch.equal(a.scores, c.scores) and torch.equal(a.pred_classes, c.pred_classes)
        assert torch.equal(a.pred_masks, c.pred_masks) and torch.equal(a.pred_plane, c.pred_plane)
        assert torch.equal(a.pred_rot_axis, c.pred_rot_axis) and torch.equal(a.pred_tran_axis, c.pred_tran_axis) and torch.equal(o["depth"], f["depth"])
    for b, o in enumerate(outs):
        inst = o["instances"]
        idx = out_b.keep[b, : int(out_b.det.count[b])].bool().nonzero().squeeze(1)
        r0 = int(out_b.det.row_offset[b])
        assert len(inst) == len(idx) > 0
        assert torch.equal(inst.pred_boxes.tensor, out_b.boxes[b, idx]) and torch.equal(inst.scores, out_b.det.scores[b, idx])
        assert inst.pred_classes.dtype == torch.int64 and inst.pred_masks.dtype == torch.bool
        assert torch.equal(inst.pred_masks, out_b.masks[b, idx].bool())
        assert torch.equal(inst.pred_plane, out_b.det.pred_plane[r0 + idx]) and torch.equal(inst.pred_rot_axis, out_b.det.pred_rot_axis[r0 + idx])
        assert torch.equal(o["depth"], out_b.depth[b])
    # PlaneRCNN_Branch.process: COCO records + plane offsets, then create_instances
    branch = PlaneRCNN_Branch.__new__(PlaneRCNN_Branch)
    branch._cpu_device, branch._refine_on = "cpu", False
    pred = branch.process(outs[0])
    idx = out_b.keep[0, : int(out_b.det.count[0])].bool().nonzero().squeeze(1)
    assert ((pred["pred_plane"] - out_b.planes[0, idx].cpu()).abs().max() / out_b.planes[0, idx].abs().max()).item() < 1e-5
    assert len(pred["instances"]) == len(idx) and set(pred["instances"][0]) == {"image_id", "category_id", "bbox", "score", "segmentation"}
    ci = create_instances(pred["instances"], HW, pred_planes=pred["pred_plane"].numpy(), pred_rot_axis=pred["pred_rot_axis"],
                          pred_tran_axis=pred["pred_tran_axis"], conf_threshold=0.3)
    k = int((outs[0]["instances"].scores > 0.3).sum())
    assert len(ci) == k and ci.pred_masks.shape == (k, 480, 640)
    assert torch.equal(ci.pred_masks.bool(), outs[0]["instances"].pred_masks[outs[0]["instances"].scores > 0.3].cpu())


def test_reference_api_odd_image_size_is_padded(hip_model, oracle, oracle_params):
    """A frame whose size is not a multiple of 32 goes through ImageList padding (planercnn.py:195) and comes back at
    its own resolution; compared stage-wise with the oracle on the same frame."""
    model, O, P = hip_model, oracle, oracle_params
    model.roi_heads.box_predictor.test_score_thresh = 0.0
    H, W = 200, 300
    frame = O.synthetic_frames(1, seed=31, h=H, w=W)[0]
    img = torch.as_tensor(frame.transpose(2, 0, 1).astype("float32"))
    out = model([{"image": img}])[0]
    inst = out["instances"]
    assert inst.image_size == (H, W) and inst.pred_masks.shape[1:] == (H, W) and len(inst) > 0
    ocfg = O.OracleCfg(score_thresh=0.0)
    x, sizes = O.preprocess([img], ocfg)
    assert tuple(x.shape[-2:]) == (224, 320) and sizes == [(H, W)]
    of = O.backbone(x, P)
    images = model.preprocess_image([{"image": img}])
    feats = model.backbone(images.tensor)
    for k in ("p2", "p4", "p6"):
        assert rel(feats[k], of[k]) < 2e-4
    assert bool((inst.pred_boxes.tensor[:, 2] <= W).all()) and bool((inst.pred_boxes.tensor[:, 3] <= H).all())
    # heads on the API's own boxes vs the oracle on the same (GPU) features
    gfe = {k: v.contiguous().cpu() for k, v in feats.items()}
    boxes = [inst.pred_boxes.tensor.cpu()]
    ref_plane = O.plane_head(O.roi_pool_fpn(gfe, boxes, *ocfg.plane_pool), P)
    assert rel(inst.pred_plane, ref_plane) < 1e-4
    # pasted masks at the odd width (byte-store path of the paste kernel) vs the oracle paste on the same probabilities
    raw = model.roi_heads.forward_with_given_boxes(feats, [_given(inst.pred_boxes.tensor, (H, W))])[0]
    ref_masks = O.paste_masks(raw.pred_masks[:, 0].cpu(), boxes[0], H, W, 0.5)
    assert (inst.pred_masks.cpu() != ref_masks).sum().item() == 0


def _given(boxes, hw):
    from articulation3d_amd.structures import Boxes, Instances

    i = Instances(hw)
    i.pred_boxes = Boxes(boxes)
    i.pred_classes = torch.zeros(len(boxes), dtype=torch.int64, device=boxes.device)
    return i


def test_no_detections_at_reference_threshold(hip_model, oracle):
    model = hip_model
    model.roi_heads.box_predictor.test_score_thresh = 0.7  # the reference default: random-init scores stay below it
    try:
        frames = oracle.synthetic_frames(2, seed=9)
        out = model.inference_batched(torch.from_numpy(frames).cuda(), want_masks=True)
        assert out.det.total == 0 and int(out.rec_count.sum()) == 0 and float(out.records.abs().sum()) == 0
        assert out.depth.shape == (2, 480, 640) and int(out.proposals[4].min()) > 100
        outs = model([{"image": torch.as_tensor(f.transpose(2, 0, 1).astype("float32"))} for f in frames])
        assert all(len(o["instances"]) == 0 for o in outs)
        assert outs[0]["instances"].pred_masks.shape == (0, 480, 640)
    finally:
        model.roi_heads.box_predictor.test_score_thresh = 0.0


def test_forward_with_given_boxes_ragged(hip_model, oracle, oracle_params, staged):
    """forward_with_given_boxes entry (roi_heads.py:147) with a different number of boxes per image, incl. none."""
    from articulation3d_amd.structures import Boxes, Instances

    model, O, P = hip_model, oracle, oracle_params
    feats = {k: v.permute(0, 3, 1, 2) for k, v in staged["feats"].items()}
    boxes = [torch.tensor([[30.0, 40, 200, 300], [100, 100, 500, 400], [5, 5, 40, 30]]), torch.zeros(0, 4)]
    insts = []
    for b in boxes:
        i = Instances(HW)
        i.pred_boxes = Boxes(b.cuda())
        i.pred_classes = torch.zeros(len(b), dtype=torch.int64).cuda()
        insts.append(i)
    from articulation3d_amd import ops as _ops
    with arithmetic(_ops, staged["precision"]):
        out = model.roi_heads.forward_with_given_boxes(feats, insts)
        torch.cuda.synchronize()
    assert out[0].pred_masks.shape == (3, 1, 28, 28) and out[1].pred_masks.shape == (0, 1, 28, 28)
    ocfg = O.OracleCfg()
    ref = O.plane_head(O.roi_pool_fpn(staged["gfeats"], boxes, *ocfg.plane_pool), P)
    assert rel(out[0].pred_plane, ref) < 1e-4 and out[1].pred_plane.shape == (0, 3)
    ora, ota = O.axis_head(O.roi_pool_fpn(staged["gfeats"], boxes, *ocfg.axis_pool), P)
    assert rel(out[0].pred_rot_axis, ora) < 1e-4 and rel(out[0].pred_tran_axis, ota) < 1e-4


@pytest.mark.parametrize("nframes", [32, 64], ids=["configs1-batch32", "configs2-clip64"])
def test_full_batch_properties(hip_model, oracle, nframes):
    """BASELINE-size batches (configs[1]: 32 frames, configs[2]: the 64-frame clip): size-independent properties instead of an
    oracle run."""
    model = hip_model
    model.roi_heads.box_predictor.test_score_thresh = 0.0
    frames = oracle.synthetic_frames(nframes, seed=21)
    fr = torch.from_numpy(frames).cuda()
    out = model.inference_batched(fr)
    out2 = model.inference_batched(fr)
    assert torch.equal(out.records, out2.records) and torch.equal(out.depth, out2.depth)  # deterministic
    # batch-size independence: frame 5 alone == frame 5 in the batch (no cross-frame leakage)
    one = model.inference_batched(fr[5:6].contiguous())
    assert torch.equal(one.records[0], out.records[5]) and torch.equal(one.depth[0], out.depth[5])
    pb, pl, _lv, _pos, pc = out.proposals
    assert int(pc.min()) > 500 and int(pc.max()) <= 1000
    assert bool((pl[:, :-1] >= pl[:, 1:]).all())  # sorted
    assert bool(((pb[..., 0] >= 0) & (pb[..., 2] <= 640) & (pb[..., 1] >= 0) & (pb[..., 3] <= 480)).all())
    n = torch.linalg.norm(out.det.pred_plane, dim=1)
    assert float((n - 1).abs().max()) < 1e-5  # unit normals
    assert float((torch.linalg.norm(out.det.pred_rot_axis[:, :2], dim=1) - 1).abs().max()) < 1e-5
    assert int(out.rec_count.sum()) == int(out.keep.sum())


def test_detect_clip_matches_reference_style_loop(hip_model, oracle):
    """pipeline.detect_clip (batched + records + re-paste) == per-frame inference -> process -> create_instances."""
    from articulation3d_amd.pipeline import detect_clip
    from articulation3d_amd.utils.arti_vis import PlaneRCNN_Branch, create_instances
    from articulation3d_amd.utils.opt_utils import track_planes

    model = hip_model
    model.roi_heads.box_predictor.test_score_thresh = 0.0
    frames = oracle.synthetic_frames(5, seed=41)
    preds = detect_clip(model, frames, batch=2, conf_threshold=0.35)
    assert len(preds) == 5
    branch = PlaneRCNN_Branch.__new__(PlaneRCNN_Branch)
    branch._cpu_device, branch._refine_on = "cpu", False
    for f, p in zip(frames, preds):
        out = model([{"image": torch.as_tensor(f.transpose(2, 0, 1).astype("float32"))}])[0]
        pd = branch.process(out)
        ref = create_instances(pd["instances"], HW, pred_planes=pd["pred_plane"].numpy(), pred_rot_axis=pd["pred_rot_axis"],
                               pred_tran_axis=pd["pred_tran_axis"], conf_threshold=0.35)
        assert len(p) == len(ref) > 0
        assert torch.equal(p.pred_boxes.tensor, ref.pred_boxes.tensor) and np.array_equal(p.pred_classes, ref.pred_classes)
        np.testing.assert_allclose(p.scores, ref.scores, rtol=0, atol=0)
        assert torch.equal(p.pred_rot_axis, ref.pred_rot_axis) and torch.equal(p.pred_tran_axis, ref.pred_tran_axis)
        assert (p.pred_planes - ref.pred_planes).abs().max() <= 1e-5 * ref.pred_planes.abs().max()
        assert torch.equal(p.pred_masks, ref.pred_masks)  # re-pasted on the receiving side == pasted by the sender
    planes = track_planes(preds)  # runs on the rebuilt records (5 frames: every track is filtered as too short)
    assert planes == {"rot": [], "trans": []}


def test_seeded_fuzz_of_every_fp32_grade_kernel(ops):
    """tools/fuzz_kernels.py with a fixed seed and a 45 s budget: random layers of every kind the detector uses, in all three fp32-grade
    arithmetics, each against a float64 convolution, and every kernel form that claims the dispatcher's bits held to them.  The
    dispatcher's own log must show the kernel families of all three modes (the coverage statement of the run)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("a3d_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_kernels.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    out = fuzz.run(seed=20260104, budget_s=45.0, verbose=False)
    assert out["cases"] >= 20, out["cases"]
    assert not out["failures"], out["failures"][:5]
    fam = {v.split("<")[0].split(" ")[0] for v in out["variants"]}
    want = {"conv_h2_kernel", "conv_h2w_kernel", "conv_h2xs_kernel", "wino_gemm_h2w_kernel",  # fp16x2
            "conv_x3_kernel", "wino_gemm_x3w_kernel",                                          # bf16x3
            "conv_gemm_v2_kernel", "conv_pw_kernel", "wino_fused_kernel", "wino_gemm_kernel"}  # fp32-input MFMA
    assert want <= fam, (want - fam, sorted(fam))


def test_seeded_fuzz_of_the_round4_kernels(ops):
    """tools/fuzz_kernels.py run_round4, fixed seed, 25 s: random image sizes through the one-launch stem (bits of the two launches), random
    layers through the transposed-read weight gradient (float64 of the bf16-rounded operands; bf16-stored == fp32-stored operands bit for
    bit), random box sets through the ROIAlign backward's tile gather (two runs bit for bit; the float-atomics form to fp32 rounding)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("a3d_fuzz4", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_kernels.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    out = fuzz.run_round4(seed=20260301, budget_s=25.0, verbose=False)
    print(out["cases"])
    assert not out["failures"], out["failures"]
    assert min(out["cases"].values()) >= 3, out["cases"]


@pytest.mark.parametrize("B", [5, 64])
def test_shared_filter_conv_over_the_pyramid_levels_as_one_launch(ops, B):
    """Round 5, SURVEY K5: the RPN head's 3x3 conv (StandardRPNHead, planercnn.py:168 -> detectron2) applies ONE filter to p2..p6.
    ops.conv2d_levels transforms every level into its slice of one V buffer and runs ONE Winograd GEMM launch over the concatenated
    tiles (a3d_wino_gemm_levels; a per-tile level table in the epilogue).  Outputs and recorded maxima equal the per-level launches
    bit for bit -- at 64 frames (every level a whole number of 128-tile blocks) and at 5 frames (blocks that straddle levels and
    images) -- and a layer that reads one level's slice afterwards (the depth head's lateral conv shares the transform) gets the
    bits of its own launch too."""
    if ops.DEFAULT_PRECISION != 3:
        pytest.skip("the fp16x2 arithmetic's form")
    torch.manual_seed(23)
    sizes = [(120, 160), (60, 80), (30, 40), (15, 20), (8, 10)] if B <= 8 else [(60, 80), (30, 40), (15, 20), (8, 10)]  # (64 frames: p3-p6 keep the test light)
    feats = [torch.randn(B, h, w, 256, device="cuda") * (1 + l) * torch.logspace(-1, 1, B, device="cuda").view(B, 1, 1, 1) for l, (h, w) in enumerate(sizes)]
    pk = ops.pack_conv(torch.randn(256, 256, 3, 3) / 48, torch.randn(256) * 0.1, None, 1, 1, ops.ACT_RELU)
    other = ops.pack_conv(torch.randn(128, 256, 3, 3) / 48, torch.randn(128) * 0.1, None, 1, 1, ops.ACT_LEAKY)
    per_level = [ops.conv2d(f, pk) for f in feats]
    other_ref = ops.conv2d(feats[1], other, wino=True)
    with ops.share_wino_input(feats):
        one = ops.conv2d_levels(feats, pk)
        assert ops.last_conv_variant() == f"wino_gemm_h2w_kernel<4> levels{len(feats)}", ops.last_conv_variant()
        co = ops.conv2d(feats[1], other, wino=True)  # reads level 1's slice of the shared buffer
    for a, b in zip(one, per_level):
        assert torch.equal(a, b) and torch.equal(a._a3d_amax, b._a3d_amax)
    assert torch.equal(co, other_ref)
    ops.WINO_LEVELS = False
    try:
        assert all(torch.equal(a, b) for a, b in zip(ops.conv2d_levels(feats, pk), per_level))
        assert ops.last_conv_variant().startswith("wino_gemm_h2w_kernel<") and "levels" not in ops.last_conv_variant()
    finally:
        ops.WINO_LEVELS = True
