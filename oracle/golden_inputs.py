"""Seeded inputs / weights of the golden fixtures (tests/golden/*.npz).  Shared by oracle/make_golden.py
(which feeds them to the REFERENCE modules in the build container) and by the tests (which feed them to the
oracle and to the HIP path), so the fixtures only need to store expected outputs.  TEST INFRASTRUCTURE."""
from __future__ import annotations

import torch


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def paste_case():
    """28x28 mask probabilities + boxes incl. out-of-image, sub-pixel, degenerate-thin and full-frame cases."""
    g = _gen(101)
    n = 8
    masks = torch.rand(n, 28, 28, generator=g)
    masks[1] = (masks[1] > 0.5).float()  # hard mask
    masks[2] = 0.5  # exactly at the threshold everywhere
    boxes = torch.tensor([
        [10.3, 20.7, 100.9, 140.2],
        [-15.5, -8.25, 60.0, 75.5],      # partly outside (top-left)
        [100.0, 60.0, 200.0, 130.0],     # integer box, threshold-valued mask
        [150.2, 100.1, 150.9, 119.4],    # sub-pixel wide
        [0.0, 0.0, 160.0, 120.0],        # full frame
        [120.5, 90.5, 190.75, 140.0],    # overshoots bottom-right
        [33.3, 44.4, 35.5, 46.6],        # ~2x2 pixels
        [80.0, 10.0, 81.0, 110.0],       # 1 pixel wide, tall
    ])
    return masks, boxes, (120, 160)


def head_params(kind: str, seed: int = 7):
    """Random parameters with the reference's state_dict names for one head (values ~ its initialisers' scale)."""
    g = _gen(seed + {"plane": 0, "axis": 1, "depth": 2}[kind])
    r = lambda *s, std=1.0: torch.randn(*s, generator=g) * std
    P = {}
    if kind == "plane":
        pre = "roi_heads.plane_head."
        for k in range(1, 5):
            P[pre + f"plane_conv{k}.weight"] = r(256, 256, 3, 3, std=0.03)
            P[pre + f"plane_conv{k}.bias"] = r(256, std=0.05)
        P[pre + "plane_fc1.weight"] = r(1024, 256 * 14 * 14, std=0.008)
        P[pre + "plane_fc1.bias"] = r(1024, std=0.05)
        P[pre + "param_pred.weight"] = r(3, 1024, std=0.03)
        P[pre + "param_pred.bias"] = r(3, std=0.05)
    elif kind == "axis":
        pre = "roi_heads.axis_head."
        for t in ("R", "T"):
            for k in range(1, 5):
                P[pre + f"axis_{t}_conv{k}.weight"] = r(256, 256, 3, 3, std=0.03)
                P[pre + f"axis_{t}_conv{k}.bias"] = r(256, std=0.05)
            P[pre + f"axis_{t}_fc1.weight"] = r(1024, 256 * 14 * 14, std=0.008)
            P[pre + f"axis_{t}_fc1.bias"] = r(1024, std=0.05)
        for nm, n in (("rotation", 2), ("offset", 1), ("translation", 2)):
            P[pre + f"{nm}.weight"] = r(n, 1024, std=0.03)
            P[pre + f"{nm}.bias"] = r(n, std=0.05)
    elif kind == "depth":
        pre = "depth_head."

        def bn(name, c):
            P[name + ".weight"] = 0.5 + torch.rand(c, generator=g)
            P[name + ".bias"] = r(c, std=0.1)
            P[name + ".running_mean"] = r(c, std=0.1)
            P[name + ".running_var"] = 0.5 + torch.rand(c, generator=g)

        for i in range(1, 6):
            P[pre + f"conv{i}.0.weight"] = r(128, 256, 3, 3, std=0.02)
            P[pre + f"conv{i}.0.bias"] = r(128, std=0.05)
            bn(pre + f"conv{i}.1", 128)
        for i, (ci, co) in enumerate(((128, 128), (256, 128), (256, 128), (256, 128), (256, 64)), start=1):
            P[pre + f"deconv{i}.1.weight"] = r(co, ci, 3, 3, std=0.03)
            P[pre + f"deconv{i}.1.bias"] = r(co, std=0.05)
            bn(pre + f"deconv{i}.2", co)
        P[pre + "depth_pred.weight"] = r(1, 64, 3, 3, std=0.05)
        P[pre + "depth_pred.bias"] = r(1, std=0.05)
    else:
        raise ValueError(kind)
    return P


def head_input(n: int = 5):
    """Pooled ROI features [n,256,14,14] (post-ReLU-like, non-negative)."""
    return torch.rand(n, 256, 14, 14, generator=_gen(202)) * 2.0


def depth_features():
    g = _gen(303)
    shapes = {"p2": (120, 160), "p3": (60, 80), "p4": (30, 40), "p5": (15, 20), "p6": (8, 10)}
    return {k: torch.randn(1, 256, h, w, generator=g) for k, (h, w) in shapes.items()}
