"""Seeded random-shape fuzz of EVERY conv / linear kernel the three fp32-grade arithmetics dispatch.

For each random layer (1x1 / 3x3 / strided / linear / transposed 2x2 / upsampled 3x3 with and without a second source / stem), in each
of the modes fp16x2 (3), bf16x3 (2) and fp32-input MFMA (0):
  * the dispatcher's choice against a float64 convolution of the same operands (first and last image), relative to the image's
    output scale -- the fp32-grade bar of tests/test_gpu_parity.py;
  * every FORM that claims the dispatcher's bits must deliver them: narrow 128 x 64 / 128 x 128 tiles (tune 10 / 11), the wide 256 x 256
    kernel (tune 9), per-lane instead of row-major epilogue (tune 12), the activation-stationary pointwise kernel on / off (tune 13 / 14), the small-grid pointwise kernel (tune 17),
    pre-split activations through the dual-DMA kernel, Winograd one-launch / plane-split, the bf16x3 narrow Winograd GEMM (tune 8), the
    fp32 one-launch against the two-launch Winograd (tune 7), the persistent pointwise kernel on / off (tune 6 / 5), fused against
    four-launch upsampled convs.  A form the launcher refuses for the shape (A3D_ERR_UNSUPPORTED) is skipped, not failed.
The set of kernel variants the dispatcher reported is returned (and printed): the coverage statement of the run.

    python tools/fuzz_kernels.py [budget_seconds] [seed]        (tests/test_gpu_parity.py runs it with a fixed seed and a 45 s budget)"""
import os
import random
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from articulation3d_amd import ops  # noqa: E402

MODES = {3: "fp16x2", 2: "bf16x3", 0: "fp32"}
TOL = 5e-6  # max |y - y64| over an image / max |y64| of that image (tests/test_gpu_parity.py's bar for one fp32-grade layer)


def _try(fn):
    try:
        return fn(), None
    except RuntimeError as e:  # a form the launcher does not offer for this shape
        if "UNSUPPORTED" in str(e) or "A3D_ERR_ARG" in str(e):
            return None, str(e)
        raise


def _ref_check(y, ref, imgs):
    got = y[imgs].double().cpu()
    scale = ref.abs().flatten(1).amax(1).clamp_min(1e-30).view(-1, *([1] * (ref.dim() - 1)))
    return float(((got - ref).abs() / scale).max())


def run(seed: int = 7, budget_s: float = 60.0, verbose: bool = True, max_cases: int = 10 ** 6):
    rng = random.Random(seed)
    t_end = time.time() + budget_s
    seen, fails, cases = set(), [], 0
    saved_mode, saved_ps = ops.DEFAULT_PRECISION, ops.WINO_PLANE_SPLIT

    def note():
        v = ops.last_conv_variant()
        seen.add(v)
        return v

    def launch(mode, x, pk, **kw):
        ops.DEFAULT_PRECISION = mode
        y = ops.conv2d(x, pk, **kw)
        return y, note()

    try:
        while time.time() < t_end and cases < max_cases:
            cases += 1
            torch.manual_seed(seed * 100003 + cases)
            kind = rng.choice(["1x1", "1x1", "3x3", "3x3", "3x3s2", "linear", "deconv", "ups", "ups2", "stem"])
            B = rng.randint(1, 5)
            H, W = rng.randint(3, 44), rng.randint(3, 52)
            big = cases == 1  # one layer large enough for the kernels only multi-round problems reach (wide bf16x3 Winograd GEMM, one-launch fp32 Winograd)
            if big:
                kind, B, H, W = "3x3", 24, 60, 80
            act = rng.choice([ops.ACT_RELU, ops.ACT_NONE, ops.ACT_LEAKY])
            spread = torch.logspace(-2, 2, B, device="cuda").view(B, 1, 1, 1)
            x2 = None
            kw = {}
            desc = kind
            if kind in ("1x1", "3x3", "3x3s2"):
                k = 1 if kind == "1x1" else 3
                st = 2 if kind == "3x3s2" or (kind == "1x1" and rng.random() < 0.25) else 1
                Cin = rng.choice([32, 64, 128, 256, 512] if k == 3 else [32, 64, 128, 256, 512, 1024, 2048])
                Cout = rng.choice([16, 36, 64, 128, 256, 512] if k == 3 else [12, 16, 64, 128, 256, 512, 1024])
                if big:
                    st, Cin, Cout = 1, 256, 256
                w = torch.randn(Cout, Cin, k, k) / (k * Cin ** 0.5)
                bn = (torch.rand(Cout) + 0.5, torch.randn(Cout) * 0.1, torch.randn(Cout) * 0.1, torch.rand(Cout) + 0.5, 1e-5) if rng.random() < 0.5 else None
                bias = None if bn is not None else torch.randn(Cout) * 0.1
                pk = ops.pack_conv(w, bias, bn, st, k // 2, act)
                x = torch.randn(B, H, W, Cin, device="cuda") * spread
                Ho, Wo = (H + 2 * (k // 2) - k) // st + 1, (W + 2 * (k // 2) - k) // st + 1
                with_res = rng.random() < 0.4 and not big  # (a residual keeps a 3x3 layer off the Winograd forms)
                if with_res:
                    kw["res"] = torch.randn(B, Ho, Wo, pk.cols, device="cuda")

                def ref_fn(imgs, w=w, bias=bias, bn=bn, st=st, k=k, kw=kw, act=act):
                    r = F.conv2d(x[imgs].double().permute(0, 3, 1, 2).cpu(), w.double(), None if bias is None else bias.double(), stride=st, padding=k // 2)
                    if bn is not None:
                        sc = bn[0].double() / torch.sqrt(bn[3].double() + bn[4])
                        r = r * sc.view(1, -1, 1, 1) + (bn[1].double() - bn[2].double() * sc).view(1, -1, 1, 1)
                    r = r.permute(0, 2, 3, 1)
                    if "res" in kw:
                        r = r + kw["res"][imgs][..., : r.shape[-1]].double().cpu()
                    return r
                desc = f"{kind} {B}x{H}x{W}x{Cin}->{Cout} s{st} res={int(with_res)} bn={int(bn is not None)}"
            elif kind == "linear":
                M, K, N = rng.randint(1, 700), rng.choice([256, 1024, 4096]), rng.choice([12, 256, 1024])
                w = torch.randn(N, K) / K ** 0.5
                bias = torch.randn(N) * 0.1
                pk = ops.pack_linear(w, bias, act=act)
                x = (torch.randn(M, K, device="cuda") * torch.logspace(-2, 2, M, device="cuda").view(M, 1)).view(M, 1, 1, K)
                B = M
                ref_fn = lambda imgs, w=w, bias=bias: (x[imgs].double().cpu().view(len(imgs), -1) @ w.double().t() + bias.double()).view(len(imgs), 1, 1, -1)
                desc = f"linear {M}x{K}->{N}"
            elif kind == "deconv":
                Cin, Cout = rng.choice([64, 256]), rng.choice([64, 256])
                w = torch.randn(Cin, Cout, 2, 2) / Cin ** 0.5
                bias = torch.randn(Cout) * 0.1
                pk = ops.pack_deconv2x2(w, bias, act)
                x = torch.randn(B, H, W, Cin, device="cuda") * spread
                ref_fn = lambda imgs, w=w, bias=bias: F.conv_transpose2d(x[imgs].double().permute(0, 3, 1, 2).cpu(), w.double(), bias.double(), stride=2).permute(0, 2, 3, 1)
                desc = f"deconv2x2 {B}x{H}x{W}x{Cin}->{Cout}"
            elif kind in ("ups", "ups2"):
                C1, Cout = rng.choice([64, 128]), rng.choice([64, 128])
                C2 = C1 if kind == "ups2" else 0
                w = torch.randn(Cout, C1 + C2, 3, 3) / (3 * (C1 + C2) ** 0.5)
                bias = torch.randn(Cout) * 0.1
                phases = ops.pack_conv_ups_phases(w, bias, None, act)
                x = torch.randn(B, H, W, C1, device="cuda") * spread
                x2 = torch.randn(B, H, W, C2, device="cuda") * spread if C2 else None

                def ref_fn(imgs, w=w, bias=bias, x2=x2):
                    xi = x[imgs] if x2 is None else torch.cat([x[imgs], x2[imgs]], -1)
                    up = F.interpolate(xi.double().permute(0, 3, 1, 2).cpu(), scale_factor=2, mode="nearest")
                    return F.conv2d(up, w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
                desc = f"ups3x3 {B}x{H}x{W}x({C1}+{C2})->{Cout}"
            else:  # stem: 7x7 s2 p3 on the NHWC4 frame tensor
                Hs, Ws = 2 * rng.randint(8, 40), 2 * rng.randint(8, 40)
                w = torch.randn(64, 3, 7, 7) / 12.0
                bn = (torch.rand(64) + 0.5, torch.randn(64) * 0.1, torch.randn(64) * 0.1, torch.rand(64) + 0.5, 1e-5)
                pk = ops.pack_stem(w, bn)
                x = torch.zeros(B, Hs, Ws, 4, device="cuda")
                x[..., :3] = torch.randn(B, Hs, Ws, 3, device="cuda") * 60
                act = ops.ACT_RELU

                def ref_fn(imgs, w=w, bn=bn):
                    r = F.conv2d(x[imgs][..., :3].double().permute(0, 3, 1, 2).cpu(), w.double(), None, stride=2, padding=3)
                    sc = bn[0].double() / torch.sqrt(bn[3].double() + bn[4])
                    return (r * sc.view(1, -1, 1, 1) + (bn[1].double() - bn[2].double() * sc).view(1, -1, 1, 1)).permute(0, 2, 3, 1)
                desc = f"stem {B}x{Hs}x{Ws}"
            imgs = sorted({0, B - 1})
            ref = ref_fn(imgs)
            if act == ops.ACT_RELU:
                ref = torch.relu(ref)
            elif act == ops.ACT_LEAKY:
                ref = F.leaky_relu(ref, 0.01)
            n = ref.shape[-1]
            for mode in (3, 2, 0):
                problems = []
                if kind in ("ups", "ups2"):
                    ops.DEFAULT_PRECISION = mode
                    y = ops.conv2d_ups(x, phases, x2=x2)
                    v0 = note()
                    forms = []
                    if mode == 3 and v0.startswith("conv_ph4p_kernel"):  # the patch-resident kernel reduces in another order: held to float64 only;
                        t15, _ = _try(lambda: ops.conv2d_ups(x, phases, x2=x2, tune=15))  # the tap-outer form claims the four-launch bits
                        note()
                        f4 = ops.conv2d_ups(x, phases, x2=x2, fused=False)
                        note()
                        if t15 is not None and not torch.equal(t15, f4):
                            problems.append("tap-outer fused form: bits differ from the four launches'")
                    else:
                        forms = [("four launches", lambda: ops.conv2d_ups(x, phases, x2=x2, fused=False))]
                else:
                    y, v0 = launch(mode, x, pk, **kw)
                    forms = []
                    plain = kind in ("1x1", "3x3", "3x3s2", "linear")
                    # (an explicit tile variant goes with an explicit arithmetic: the mode rules only pick 2 / 3 for tune 0)
                    own = (mode == 3 and v0.startswith(("conv_h2", "conv_c3p"))) or (mode == 2 and v0.startswith("conv_x3"))
                    if own and plain:
                        forms += [(f"tune {t}", lambda t=t: launch(mode, x, pk, tune=t, precision=mode, **kw)[0]) for t in (9, 10, 11, 12)]
                    if own and mode == 3 and kind == "1x1":
                        forms += [(f"tune {t}", lambda t=t: launch(3, x, pk, tune=t, precision=3, **kw)[0]) for t in (13, 14, 17)]  # (17: the small-grid form)
                    if own and mode == 3 and plain and x.shape[-1] % 16 == 0:
                        forms.append(("pre-split activations", lambda: launch(3, ops.presplit_f16x2(x), pk, **kw)[0]))
                    if mode == 3 and v0.startswith("wino"):
                        def other_ps():
                            ops.WINO_PLANE_SPLIT = not saved_ps
                            try:
                                return launch(3, x, pk, **kw)[0]
                            finally:
                                ops.WINO_PLANE_SPLIT = saved_ps
                        forms.append(("plane-split toggled", other_ps))
                        # (round 5) the ping-pong loop is the default: the lockstep loop and the 64-tile form claim its bits
                        forms += [(f"tune {t}", lambda t=t: launch(3, x, pk, tune=t, precision=3, **kw)[0]) for t in (23, 24)]
                    if mode == 2 and v0.startswith("wino"):
                        # (round 5) default = both operands pre-split by DMA, ping-pong; 8 / 24 / 25 = the register-staged narrow / 64-tile / 128-tile forms
                        forms += [(f"tune {t}", lambda t=t: launch(2, x, pk, tune=t, precision=2, **kw)[0]) for t in (8, 24, 25)]
                    if mode == 0 and v0.startswith("wino"):
                        forms.append(("tune 7", lambda: launch(0, x, pk, tune=7, **kw)[0]))
                    if mode == 0 and kind in ("1x1", "linear"):
                        forms += [(f"tune {t}", lambda t=t: launch(0, x, pk, tune=t, **kw)[0]) for t in (5, 6)]
                torch.cuda.synchronize()
                err = _ref_check(y[..., :n], ref, imgs)
                if not (err < TOL) or not bool(torch.isfinite(y).all()):
                    problems.append(f"error {err:.2e} vs float64")
                base = y
                if v0.startswith("conv_c3p"):  # the patch-resident 3x3 kernel reduces over (chunk, tap): float64 is its yardstick; the
                    base = None                # tap-outer forms are held to EACH OTHER's bits (the first one that runs is the base)
                for name, fn in forms:
                    alt, why = _try(fn)
                    note()
                    if alt is None:
                        continue
                    if base is None:
                        base = alt
                        if float(((alt - y).abs().flatten(1).amax(1) / y.abs().flatten(1).amax(1).clamp_min(1e-30)).max()) > TOL:  # (two fp32-grade sums)
                            problems.append(f"{name}: more than fp32 rounding away from the patch-resident kernel")
                        continue
                    if not torch.equal(alt, base):
                        problems.append(f"{name}: bits differ from the base form ({float((alt - base).abs().max()):.2e})")
                if verbose:
                    print(f"[{MODES[mode]:6s}] {desc}: {v0}  err {err:.1e}  {'ok' if not problems else 'FAIL ' + '; '.join(problems)}", flush=True)
                if problems:
                    fails.append((MODES[mode], desc, v0, problems))
    finally:
        ops.DEFAULT_PRECISION, ops.WINO_PLANE_SPLIT = saved_mode, saved_ps
    return dict(cases=cases, failures=fails, variants=sorted(seen))


def run_round4(seed: int = 11, budget_s: float = 30.0, verbose: bool = True):
    """Random-shape fuzz of the three kernels round 4 added outside ops.conv2d's dispatcher:
      * the one-launch stem (a3d_stem_conv_pool) against the conv launch + the pool launch: the same bits, any image size;
      * the transposed-read weight gradient (conv_wgrad_tr.hip) of the bf16 step against a float64 gradient of the bf16-rounded
        operands, and bf16-stored operands against fp32-stored ones holding the same values: the same bits;
      * the ROIAlign backward as a tile gather against the float-atomics form (fp32 rounding of the sums), two runs of it bit for bit."""
    from articulation3d_amd import train_ops as T

    rng = random.Random(seed)
    t_end = time.time() + budget_s
    fails, cases = [], {"stem": 0, "wgrad": 0, "roi": 0, "dot": 0}
    saved = ops.DEFAULT_PRECISION
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    try:
        while time.time() < t_end:
            kind = rng.choice(["stem", "wgrad", "wgrad", "roi", "dot"])
            torch.manual_seed(rng.randrange(1 << 30))
            if kind == "stem":
                ops.DEFAULT_PRECISION = 3
                B, H, W = rng.randint(1, 3), rng.randint(9, 300), rng.randint(9, 400)
                x4 = torch.randn(B, H, W, 4, device="cuda") * rng.choice([0.01, 1.0, 300.0])
                x4[..., 3] = 0
                w = torch.randn(64, 3, 7, 7) / 12
                bn = (torch.rand(64) + 0.5, torch.randn(64) * 0.1, torch.randn(64) * 0.1, torch.rand(64) + 0.5, 1e-5)
                pk = ops.pack_stem(w, bn)
                two = ops.maxpool3x3s2(ops.conv2d(x4, pk))
                one = ops.stem_pool(x4, pk)
                ok = one is not None and torch.equal(one, two) and torch.equal(ops.amax_of(one), ops.amax_of(two))
                desc = f"stem {B}x{H}x{W}"
            elif kind == "dot":  # deconv5 + depth_pred as tap products against the two layers (float64 of the same 64-channel tensor)
                ops.DEFAULT_PRECISION = 3
                B, H, W = rng.randint(1, 3), rng.randint(3, 70), rng.randint(3, 90)
                C1 = 32 * rng.randint(1, 4)
                two_src = rng.random() < 0.7
                a = torch.randn(B, H, W, C1, device="cuda") * rng.choice([0.05, 1.0, 40.0])
                c2 = torch.randn(B, H, W, C1, device="cuda") if two_src else None
                phases = ops.pack_conv_ups_phases(torch.randn(64, C1 * (2 if two_src else 1), 3, 3) / (3 * (2 * C1) ** 0.5), torch.randn(64) * 0.1, None, ops.ACT_RELU)
                w9 = (torch.randn(3, 3, 64) / 24).cuda()
                x = ops.conv2d_ups(a, phases, x2=c2)
                two = ops.conv3x3_to1(x, w9, 0.25)
                one = ops.conv2d_ups_to1(a, phases, w9, 0.25, x2=c2)
                ref = F.conv2d(x.double().permute(0, 3, 1, 2), w9.double().permute(2, 0, 1)[None], torch.tensor([0.25], dtype=torch.float64, device="cuda"), 1, 1)[:, 0]
                sc = float(ref.abs().max().clamp_min(1e-30))
                e1 = float((one.double() - ref).abs().max()) / sc if one is not None else 1.0
                e2 = float((two.double() - ref).abs().max()) / sc
                ok = one is not None and e1 < 1e-6 and e1 < 3 * e2 + 2e-7
                desc = f"dot {B}x{H}x{W}x{C1}{'x2' if two_src else ''} err {e1:.1e} | {e2:.1e}"
            elif kind == "wgrad":
                k = rng.choice([1, 1, 3])
                B, H, W = rng.randint(1, 4), rng.randint(1, 70), rng.randint(1, 90)
                if rng.random() < 0.2:
                    B, H, W = rng.randint(100, 3000), 1, 1  # linear layers
                Cin, Cout = 8 * rng.randint(1, 80), 8 * rng.randint(1, 48)
                x, dy = torch.randn(B, Cin, H, W), torch.randn(B, Cout, H, W)
                xb, dyb = x.bfloat16(), dy.bfloat16()
                wd = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
                F.conv2d(xb.double(), wd, None, 1, k // 2).backward(dyb.double())
                ref = wd.grad.permute(0, 2, 3, 1).reshape(Cout, -1)
                outs = []
                for xa, dya in ((nhwc(x).cuda(), nhwc(dy).cuda()), (nhwc(xb).cuda(), nhwc(dyb).cuda()), (nhwc(xb.float()).cuda(), nhwc(dyb).cuda())):
                    dw = torch.empty((Cout, k * k * Cin), device="cuda")
                    T.conv_wgrad(xa, dya, dw, KH=k, KW=k, stride=1, pad=k // 2, precision=1)
                    outs.append(dw)
                err = float((outs[0].double().cpu() - ref).norm() / ref.norm().clamp_min(1e-30))
                ok = err < 2e-6 and torch.equal(outs[1], outs[2]) and float((outs[1].double().cpu() - ref).norm() / ref.norm().clamp_min(1e-30)) < 2e-6
                desc = f"wgrad {B}x{H}x{W}x{Cin}->{Cout} k{k} err {err:.1e}"
            else:
                B, R, C = rng.randint(1, 3), rng.randint(1, 70), rng.choice([32, 64, 256])
                Himg, Wimg = 32 * rng.randint(4, 15), 32 * rng.randint(4, 20)
                wh = torch.rand(B, R, 2) * torch.tensor([float(Wimg), float(Himg)]) * rng.choice([0.1, 0.5, 1.0]) + 2
                xy = torch.rand(B, R, 2) * torch.tensor([float(Wimg), float(Himg)]) - wh * 0.3
                boxes = torch.cat([xy, xy + wh], 2).cuda().contiguous()
                count = torch.tensor([rng.randint(0, R) for _ in range(B)], dtype=torch.int32, device="cuda")
                off = torch.zeros(B, dtype=torch.int32, device="cuda")
                off[1:] = torch.cumsum(count, 0)[:-1].to(torch.int32)
                rows = int(count.sum())
                dout = torch.randn(max(rows, 1), 7, 7, C, device="cuda")
                mk = lambda: [torch.zeros(B, Himg // s_, Wimg // s_, C, device="cuda") for s_ in (4, 8, 16, 32)]
                a, a2, sc = mk(), mk(), mk()
                args = ([1 / 4, 1 / 8, 1 / 16, 1 / 32], boxes, dout)
                T.roi_align_fpn_backward(a, *args, P=7, sampling_ratio=0, aligned=True, count=count, row_offset=off)
                T.roi_align_fpn_backward(a2, *args, P=7, sampling_ratio=0, aligned=True, count=count, row_offset=off)
                T.roi_align_fpn_backward(sc, *args, P=7, sampling_ratio=0, aligned=True, count=count, row_offset=off, scatter=True)
                nrm = max(float(torch.cat([t.flatten() for t in sc]).norm()), 1e-30)
                err = float(torch.cat([(p - q).flatten() for p, q in zip(a, sc)]).norm()) / nrm
                ok = all(torch.equal(p, q) for p, q in zip(a, a2)) and err < 2e-6
                desc = f"roi bwd {B}x{R} C{C} {Himg}x{Wimg} rows {rows} err {err:.1e}"
            cases[kind] += 1
            if verbose:
                print(f"{desc}: {'ok' if ok else 'FAIL'}", flush=True)
            if not ok:
                fails.append(desc)
    finally:
        ops.DEFAULT_PRECISION = saved
    return dict(cases=cases, failures=fails)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "round4":
        o4 = run_round4(seed=int(sys.argv[3]) if len(sys.argv) > 3 else 11, budget_s=float(sys.argv[2]) if len(sys.argv) > 2 else 30.0)
        print(o4["cases"], "FAILURES:", o4["failures"])
        sys.exit(1 if o4["failures"] else 0)
    out = run(seed=int(sys.argv[2]) if len(sys.argv) > 2 else 7, budget_s=float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
    print(f"\n{out['cases']} random layers x 3 arithmetics; kernel variants the dispatcher reported ({len(out['variants'])}):")
    for v in out["variants"]:
        print("  ", v)
    print("FAILURES:", len(out["failures"]))
    for f in out["failures"]:
        print("  ", f)
    sys.exit(1 if out["failures"] else 0)
