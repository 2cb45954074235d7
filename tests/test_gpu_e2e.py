"""GPU suite (-m gpu): the north star's literal acceptance criterion -- MATCHED DETECTIONS, end to end.

Frames in -> detections out through `PlaneRCNN.inference_batched` (HIP kernels behind the C ABI) and through
`planercnn_oracle.detect` (the CPU restatement of pkg/modeling/meta_arch/planercnn.py:125-184 + pkg/utils/arti_vis.py:54-149),
each path running on ITS OWN upstream tensors: nothing is shared between the two but the uint8 frames and the weights.
Per frame, at SCORE_THRESH_TEST 0.5 (a handful of detections) and 0.0 (the 100-detection cap):

  DISCRETE (asserted exactly): equal detection count; the same detections rank for rank -- every oracle detection pairs with a
    HIP detection of the same class whose box is within 5e-3 px, ranks exchanged only between scores tied to 2e-4 (the rank
    order is defined by a score that itself carries 6e-5 of end-to-end rounding);
  CONTINUOUS (asserted against a yardstick): boxes <= 5e-3 px and scores <= 1e-4 of the oracle's; plane normals, rotation /
    translation axis parameters, plane normal*offset, depth:  | HIP - float64 |  <=  max(1e-4, 3 x | oracle fp32 - float64 |),
    where float64 is the exact evaluation of the same graph with the oracle's discrete choices imposed (oracle/exact.py).
    The per-ROI head outputs are L2-NORMALISED 2- / 3-vectors n = r / |r|, whose error is the raw vector's divided by |r|.  Round 3
    compares what the arithmetic actually produces, at its MAXIMUM over all detections, no quantiles and no gross-error escape:
      - the RAW vectors r (param_pred, rotation | offset, translation FC outputs before F.normalize; the heads' keep_raw hook), and
      - every normalised output weighted by its own conditioning, | n_hip - n_64 | * | r_64 |,
    each <= max(floor, 3 x the same statistic of the oracle's fp32 evaluation).
    Why a yardstick and not a flat 1e-4: end to end, two fp32 evaluations of this graph differ by 3e-5 .. 9e-5 at the FPN outputs
    (CPU summation orders among themselves -- oracle/seed_search.py -- and HIP vs CPU alike), and the 6-layer heads that end in an
    L2-normalised 2- or 3-vector amplify that to 1e-4 .. 4e-3 on random-init weights.  The flat 1e-4 (BASELINE.json north_star)
    holds, and is asserted, where both paths get IDENTICAL inputs: tests/test_gpu_parity.py::test_stage_roi_heads_paste_lsq_and_records.
  REPORTED: pasted-mask Hamming distance (pixels within rounding of the 0.5 threshold may flip; bit-exact paste on identical
    inputs is asserted in test_gpu_parity.py), and the margin of every discrete decision of the oracle run to its threshold / tie.

The frames are the seeds of tests/golden/e2e_frames.json, chosen by the committed oracle-only search oracle/seed_search.py
(detections invariant under three alternative fp32 evaluation orders of the backbone on the CPU).  The report is printed
(`pytest -s`) and always written to gpurun_out/e2e_matched_t<thresh>.json.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONT = ("box_err_px", "score_err", "plane_rel", "rot_axis_rel", "tran_axis_rel", "plane_offset_rel", "depth_rel")
K_YARD = 3.0
RAW_FLOOR = 2e-5  # of the largest raw component: below this the comparison is fp32 rounding of the final 1024-deep FC itself


def _frames(golden_dir, oracle, n=8):
    doc = json.load(open(os.path.join(golden_dir, "e2e_frames.json")))
    sel = doc["frames"][:n]
    assert len(sel) == n
    frames = np.concatenate([oracle.synthetic_frames(1, seed=f["seed"]) for f in sel])
    return sel, frames


_ORACLE_RUNS = {}


def _oracle_runs(thresh, golden_dir, O, P):
    """fp32 oracle + float64 exact evaluation of the committed frames at one threshold: computed once, shared by the three
    arithmetic modes the HIP path is run in."""
    if thresh not in _ORACLE_RUNS:
        from oracle import exact as E
        from oracle import matching as M

        sel, frames = _frames(golden_dir, O)
        ocfg = O.OracleCfg(score_thresh=thresh)
        imgs = O.frames_to_chw(frames)
        P64 = E.to_double(P)
        o32s, o64s = [], []
        for i in range(len(imgs)):  # one frame at a time: bounds the float64 run's memory
            o32, aux = O.detect(imgs[i:i + 1], P, ocfg, return_aux=True)
            o64 = E.detect_exact(imgs[i:i + 1], P64, ocfg, o32, aux)
            o32s.append(o32[0])
            o64s.append(o64[0])
        cpu64 = [M.compare_frame(a, b) for a, b in zip(o32s, o64s)]
        _ORACLE_RUNS[thresh] = (sel, frames, o32s, o64s, cpu64)
    return _ORACLE_RUNS[thresh]


MODES = {3: "fp16x2", 2: "bf16x3", 0: "fp32"}


@pytest.mark.parametrize("precision", [3, 2, 0], ids=lambda p: MODES[p])
@pytest.mark.parametrize("thresh", [0.5, 0.0])
def test_matched_detections_end_to_end(hip_model, oracle, oracle_params, golden_dir, thresh, precision):
    """The acceptance criterion in every arithmetic the library offers as fp32-grade: the default fp16x2 (precision 3), bf16x3 (2)
    and the fp32-input MFMA (0) -- the `alt_modes` of the bench line are backed by the same assertions as the headline."""
    from articulation3d_amd import ops
    from oracle import matching as M

    O, P, model = oracle, oracle_params, hip_model
    sel, frames, o32s, o64s, cpu64 = _oracle_runs(thresh, golden_dir, O, P)
    model.roi_heads.box_predictor.test_score_thresh = thresh
    saved = ops.DEFAULT_PRECISION
    ops.DEFAULT_PRECISION = precision
    model.roi_heads.plane_head.keep_raw = model.roi_heads.axis_head.keep_raw = True
    window0 = ops.roi_window_count() if precision == 3 else 0
    try:
        out = model.inference_batched(torch.from_numpy(frames).cuda(), want_masks=True)
        torch.cuda.synchronize()
        got = M.gpu_frame_results(out)
    finally:
        model.roi_heads.box_predictor.test_score_thresh = 0.0
        model.roi_heads.plane_head.keep_raw = model.roi_heads.axis_head.keep_raw = False
        ops.DEFAULT_PRECISION = saved
    if precision == 3:  # the window monitor of the default arithmetic: no ROI of these frames is fainter than 2^-16 of its level
        assert ops.roi_window_count() == window0
    vs32 = [M.compare_frame(g, o) for g, o in zip(got, o32s)]
    hip64 = [M.compare_frame(g, o) for g, o in zip(got, o64s)]
    s32, sh, sc = M.summarize(vs32), M.summarize(hip64), M.summarize(cpu64)
    margins = [f["margins"][str(thresh)] for f in sel]
    report = dict(score_thresh_test=thresh, arithmetic=MODES[precision], seeds=[f["seed"] for f in sel], hip_vs_oracle_fp32=s32, hip_vs_float64=sh,
                  oracle_fp32_vs_float64=sc,
                  min_margins={k: min(m[k] for m in margins) for k in margins[0] if k != "detections"})
    print("\nmatched detections @ thresh", thresh, MODES[precision], json.dumps(report, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    tag = "" if precision == 3 else "_" + MODES[precision]
    with open(os.path.join(ROOT, "gpurun_out", f"e2e_matched_t{thresh}{tag}.json"), "w") as f:
        json.dump(dict(report=report, frames_vs_fp32=vs32, frames_hip_vs_f64=hip64, frames_cpu_vs_f64=cpu64), f, indent=1)
    for s, m in zip(report["seeds"], vs32):  # discrete identity + boxes / scores, frame by frame
        assert m["same_count"], (s, m)
        assert m["classes_equal"], (s, m)
        assert m["box_err_px"] <= M.TOL["box_px"] and m["score_err"] <= M.TOL["score"] and m["matched"], (s, m)
    # The yardstick: as close to the exact (float64) evaluation as the reference's own fp32 arithmetic is.  Dense quantities by their
    # maximum.
    for k in ("plane_offset_rel", "depth_rel"):
        hip, cpu = sh["max_" + k], sc["max_" + k]
        assert hip <= max(1e-4, K_YARD * cpu), (k, hip, cpu)
    # Per-ROI head outputs, at their MAXIMUM over all detections (57 at threshold 0.5, 800 at 0.0):
    #  (1) the raw vectors before F.normalize -- no amplification, so a plain maximum is a fair statistic;
    #  (2) every normalised output with its conditioning divided out: |n_hip - n_64| * |r_64|  (n = r / |r|  =>  dn ~ dr / |r|).
    # The quantile rule and the 10x gross-error clause of round 2 are gone: the heavy tail they existed for is exactly the factor
    # 1 / |r|, and (2) removes it detection by detection.
    for k in ("raw_plane_err", "raw_rot_err", "raw_tran_err", "plane_cond", "rot_axis_cond", "tran_axis_cond"):
        hip, cpu = sh["max_" + k], sc["max_" + k]
        assert hip <= max(RAW_FLOOR, K_YARD * cpu), (k, hip, cpu)
    # (equivalently: every normalised output is within  bound / |r_64|  of the exact one -- its own conditioning, nothing more)
    # ... and a gross-error bound on the SHIPPED outputs themselves (the L2-normalised plane normal and axes, at their maximum over all
    # detections): the conditioned statistics above divide the amplification 1 / |r| out, so by themselves they would let a detection
    # with a short raw vector carry an arbitrarily wrong normal.  Measured ratios HIP : CPU fp32 are 0.6-3.9.
    for k in ("plane_rel", "rot_axis_rel", "tran_axis_rel"):
        hip, cpu = sh["max_" + k], sc["max_" + k]
        assert hip <= max(1e-4, 10.0 * cpu), (k, hip, cpu)
    assert sh["mask_hamming_px"] <= K_YARD * sc["mask_hamming_px"] + 32, (sh["mask_hamming_px"], sc["mask_hamming_px"])
    if thresh == 0.0:
        assert all(d == 100 for d in s32["detections"])
    else:
        assert sum(s32["detections"]) > 0


# ------------------------------------------------------------------------------------------ the flat 1e-4, end to end, on well-conditioned heads
WELL_GAIN = float(os.environ.get("A3D_E2E_WELL_GAIN", "0.1"))  # rms of the data-dependent part W.h against a bias of norm 1 (see below)
WELL_GAIN_WIDE = 0.25  # reported beside it, held to 3e-4: the measured law is  error ~ 6e-4 x gain  (see the test's docstring)
_HEAD_OUT = (("roi_heads.plane_head.param_pred", "raw_plane", 0, 3), ("roi_heads.axis_head.rotation", "raw_rot", 0, 2),
             ("roi_heads.axis_head.offset", "raw_rot", 2, 3), ("roi_heads.axis_head.translation", "raw_tran", 0, 2))  # (layer, raw key, columns)


def _well_conditioned_params(O, P, frames, thresh, gain):
    """The last linear layer of each per-ROI head, rescaled BY VALUE for both paths: bias = a seeded vector of norm 1, weight scaled so that
    the data-dependent part W.h has rms `gain` over the detections of these frames (pass 1 of the oracle supplies that rms; the
    un-normalised axis offset gets the same treatment).  With the
    random-init layers a raw head vector r = W.h is a sum of 1024 random-sign terms around zero -- |r| down to 1e-3 of the frame's largest,
    and the shipped n = r / |r| amplifies any rounding by 1 / |r|; a trained regressor of unit normals produces |r| ~ 1.  Here every |r|
    stays >= ~0.5 while the direction still moves with the ROI (rms `gain` radians)."""
    ocfg = O.OracleCfg(score_thresh=thresh)
    imgs = O.frames_to_chw(frames)
    first = [O.detect(imgs[i:i + 1], P, ocfg)[0] for i in range(len(imgs))]
    P2 = dict(P)
    g = torch.Generator().manual_seed(606)
    for name, raw, c0, c1 in _HEAD_OUT:
        r = torch.cat([f[raw][:, c0:c1] for f in first])
        rms = float(r.pow(2).mean().sqrt())
        b = torch.randn(c1 - c0, generator=g)
        P2[name + ".weight"] = P[name + ".weight"] * (gain / rms)
        P2[name + ".bias"] = (b / b.norm()).to(P[name + ".bias"].dtype)
    return P2


def _run_well(model, O, P, frames, thresh, gain, precision):
    from articulation3d_amd import ops
    from oracle import matching as M

    P2 = _well_conditioned_params(O, P, frames, thresh, gain)
    ocfg = O.OracleCfg(score_thresh=thresh)
    imgs = O.frames_to_chw(frames)
    o32 = [O.detect(imgs[i:i + 1], P2, ocfg)[0] for i in range(len(imgs))]
    keys = [n + s for n, _, _, _ in _HEAD_OUT for s in (".weight", ".bias")]
    saved = ops.DEFAULT_PRECISION
    ops.DEFAULT_PRECISION = precision
    model.roi_heads.box_predictor.test_score_thresh = thresh
    model.load_state_dict({k: P2[k] for k in keys}, strict=False)
    model.roi_heads.plane_head.keep_raw = model.roi_heads.axis_head.keep_raw = True
    try:
        out = model.inference_batched(torch.from_numpy(frames).cuda(), want_masks=True)
        torch.cuda.synchronize()
        got = M.gpu_frame_results(out)
    finally:
        model.load_state_dict({k: P[k] for k in keys}, strict=False)
        model.roi_heads.box_predictor.test_score_thresh = 0.0
        model.roi_heads.plane_head.keep_raw = model.roi_heads.axis_head.keep_raw = False
        ops.DEFAULT_PRECISION = saved
    vs32 = [M.compare_frame(g, o) for g, o in zip(got, o32)]
    s32 = M.summarize(vs32)
    norms = {name.rsplit(".", 1)[1]: [float(v) for f in o32 for v in f[raw][:, c0:c1].norm(dim=1)] for name, raw, c0, c1 in _HEAD_OUT}
    report = dict(arithmetic=MODES[precision], gain=gain, detections=s32["detections"],
                  min_raw_norm={k: min(v) for k, v in norms.items() if v}, max_raw_norm={k: max(v) for k, v in norms.items() if v},
                  hip_vs_oracle_fp32={k: v for k, v in s32.items() if k.startswith("max_")})
    return vs32, s32, norms, report


@pytest.mark.parametrize("precision", [3, 2, 0], ids=lambda p: MODES[p])
def test_flat_1e4_on_well_conditioned_heads_end_to_end(hip_model, oracle, oracle_params, golden_dir, precision):
    """`north_star`'s criterion as written -- plane normals and axis parameters within 1e-4 relative of the reference
    path's -- stated END TO END (each path on its own upstream tensors) against the fp32 oracle, in the regime where the criterion is
    meaningful: raw head vectors of norm ~1 (see _well_conditioned_params; plane_head.py:80-82, axis_head.py:106,120).  The random-init
    case above keeps the float64 yardstick, because there the reference's own fp32 arithmetic is 2e-4 .. 1.5e-3 from exact.

    What decides the figure (measured, MI355X, fp16x2: gain 0.05 | 0.1 | 0.25 -> 2.4e-5 | ~6e-5 | 1.5e-4 on the rotation axis): the
    data-dependent part W.h of a raw vector differs between the two fp32 evaluations by ~6e-4 of ITS OWN rms -- the end-to-end
    rounding of the 1024 hidden features behind six random-init layers, the same between two CPU summation orders
    (oracle/seed_search.py) -- so the shipped unit vector moves by ~6e-4 x gain.  The flat 1e-4 therefore holds, and is asserted, up to
    a direction spread of ~0.1 rad rms across ROIs around the bias direction; at 0.25 the same law gives 1.5e-4 (asserted <= 3e-4 and
    reported).  It is a property of the graph and of fp32, not of which of the three arithmetics runs."""
    from oracle import matching as M  # noqa: F401

    O, P, model = oracle, oracle_params, hip_model
    sel, frames = _frames(golden_dir, O)
    reports = {}
    for gain, bound in ((WELL_GAIN, 1e-4), (WELL_GAIN_WIDE, 3e-4)):
        vs32, s32, norms, report = _run_well(model, O, P, frames, 0.5, gain, precision)
        reports[str(gain)] = report
        print("\nwell-conditioned heads, end to end", json.dumps(report, indent=1))
        assert sum(s32["detections"]) >= 20 and all(m["matched"] for m in vs32), s32
        assert min(min(v) for v in norms.values() if v) >= 0.4, report["min_raw_norm"]  # (the regime the test is about)
        for k in ("plane_rel", "rot_axis_rel", "tran_axis_rel"):  # the flat criterion of BASELINE.json north_star, on the shipped outputs
            assert s32["max_" + k] <= bound, (k, gain, s32["max_" + k], report)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"e2e_well_conditioned_{MODES[precision]}.json"), "w") as f:
        json.dump(reports, f, indent=1)
