"""GPU suite (-m gpu), round 4: the load-time precision audit (PlaneRCNN.audit_precision, ops.PrecisionAudit).

The default arithmetic (fp16x2) has a window: 22 significand bits under one power-of-two exponent per image.  The audit shadows every
fp16x2 layer with its bf16x3 evaluation on calibration frames and pins the layers that leave the window -- statically, per layer."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def detector():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from bench import build_detector
    from articulation3d_amd import ops

    if ops.DEFAULT_PRECISION != 3:
        pytest.skip("the audit guards the default (fp16x2) arithmetic")
    model, _ = build_detector(0.5, "cuda:0")
    return model


def _frames(n, seed=2020):
    from articulation3d_amd.utils.synthetic import synthetic_frames

    return torch.from_numpy(synthetic_frames(n, seed)).cuda()


def test_audit_is_clean_on_calibrated_weights_and_changes_nothing(detector):
    """Random-init weights with calibrated batch norm sit well inside the window: every fp16x2 layer launch satisfies the one-term law
    (worst ratio well below 1), nothing is pinned, and a detection pass after the audit gives the bits of a pass before it."""
    model = detector
    fr = _frames(2)
    before = model.inference_batched(fr)
    audit = model.audit_precision(fr)
    assert len(audit.rows) > 80, len(audit.rows)  # backbone + FPN + RPN + box head + heads + decoder
    worst = max(audit.rows, key=lambda r: r["max_ratio"])
    assert worst["max_ratio"] < 1.0 and not audit.pinned() and model.pinned_layers() == [], worst
    kinds = {r["kernel"].split("<")[0].split(" ")[0] for r in audit.rows}
    assert {"conv_h2_kernel", "wino_gemm_h2w_kernel", "conv_h2w_kernel"} <= kinds, kinds
    after = model.inference_batched(fr)
    assert torch.equal(before.records, after.records) and torch.equal(before.depth, after.depth)


def test_audit_pins_exactly_the_consumers_of_an_out_of_window_tensor(detector):
    """One batch-norm channel of res2's first block scaled by 2^28, and that channel's taps zeroed in the block's 3x3 conv2: the tensor
    conv1 produces has one channel 2^28 above the rest, so conv2 -- whose outputs do not depend on the hot channel at all -- sees 63
    live input channels 2^28 below their image's maximum, far outside the format's window, while every other layer of the detector
    still sees well-conditioned inputs.  The audit pins exactly that layer; with the pin applied a second audit finds no violation; the
    pin survives a re-pack of the layer's weights; and the pinned model's outputs match the all-bf16x3 model's (the damage is repaired,
    not just detected)."""
    from articulation3d_amd import ops

    model = detector
    blk = model.backbone.bottom_up.res2[0]
    norm = blk.conv1.norm
    w0, b0, c2w = norm.weight.clone(), norm.bias.clone(), blk.conv2.weight.detach().clone()
    fr = _frames(2, seed=77)
    try:
        with torch.no_grad():
            norm.weight[5] *= 2.0 ** 28
            norm.bias[5] = norm.bias[5].abs() * 2.0 ** 28 + 2.0 ** 28  # (positive after the ReLU on every pixel)
            blk.conv2.weight[:, 5] = 0.0
        audit = model.audit_precision(fr)
        pinned = model.pinned_layers()
        assert pinned == ["backbone.bottom_up.res2.0.conv2"], (pinned, [r for r in audit.rows if r["violations"]][:4])
        bad = [r for r in audit.rows if r["violations"]]
        assert len(bad) == 1 and bad[0]["layer"] == pinned[0] and bad[0]["max_ratio"] > 1.0
        again = model.audit_precision(fr)
        assert not [r for r in again.rows if r["violations"]], [r for r in again.rows if r["violations"]][:3]
        assert "backbone.bottom_up.res2.0.conv2" not in {r["layer"] for r in again.rows}  # (it no longer runs fp16x2: nothing to shadow)
        # the pin is a property of the layer holder: a re-pack (parameter update) keeps it
        with torch.no_grad():
            blk.conv2.weight.mul_(1.0)
        assert blk.conv2.packed().pin_precision == 2
        # repaired, not just detected: pinned fp16x2 model vs the all-bf16x3 model on the same frames
        out_p = model.inference_batched(fr)
        saved, ops.DEFAULT_PRECISION = ops.DEFAULT_PRECISION, 2
        try:
            out_x = model.inference_batched(fr)
        finally:
            ops.DEFAULT_PRECISION = saved
        rel = float((out_p.depth - out_x.depth).abs().max() / out_x.depth.abs().max())
        assert rel < 2e-4, rel
        assert torch.equal(out_p.det.count, out_x.det.count)
    finally:
        with torch.no_grad():
            norm.weight.copy_(w0)
            norm.bias.copy_(b0)
            blk.conv2.weight.copy_(c2w)
        for _n, m in model._packables():
            m.pin_precision = None
    # unpinned again, the clean weights audit clean
    assert not model.audit_precision(_frames(1)).pinned()
