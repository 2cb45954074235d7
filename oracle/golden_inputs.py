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


def axis_cases():
    """Inputs of the axis <-> (angle, offset) fixtures (SURVEY 8c fixture 4): pixel axes [n,4] (x1,y1,x2,y2) + box centres [n,2] for
    axis_to_angle_offset; (sin, cos, offset/100) triples + centres for angle_offset_to_axis; (y, x, angle) triples for
    get_boundary_point (planercnn_transforms.py:31-68,101-176).  Random in-image axes plus the edge cases the functions branch on:
    horizontal, vertical, through the centre (C = 0), endpoints outside the image, lines that miss the image, lines through corners."""
    import numpy as np

    rng = np.random.default_rng(2020)
    n = 24
    axes = np.stack([rng.uniform(0, 640, n), rng.uniform(0, 480, n), rng.uniform(0, 640, n), rng.uniform(0, 480, n)], 1)
    centers = np.stack([rng.uniform(40, 600, n), rng.uniform(40, 440, n)], 1)
    special = np.array([
        [100, 200, 500, 200],   # horizontal
        [320, 30, 320, 450],    # vertical
        [-50, -20, 700, 520],   # endpoints outside the image
        [100, 100, 300, 300],   # through its centre (C = 0: sign(C) = 0)
        [10, 470, 630, 5],
        [200, 240, 201, 240.5],  # short
        [0, 0, 639, 479],       # corner to corner
        [639, 0, 0, 479],
    ], dtype=np.float64)
    sc = np.array([[320, 240], [300, 240], [320, 240], [200, 200], [100, 100], [500, 400], [320, 240], [10, 10]], dtype=np.float64)
    axes = np.concatenate([axes, special]).astype(np.float32)
    centers = np.concatenate([centers, sc]).astype(np.float32)
    m = 24
    th = rng.uniform(-np.pi, np.pi, m)
    ao = np.stack([np.sin(th), np.cos(th), rng.uniform(0, 3.0, m)], 1)
    ao_special = np.array([
        [0.0, 1.0, 0.5],     # sin == 0: vertical line right of the centre
        [0.0, -1.0, 0.5],    # ... left of it
        [1.0, 0.0, 0.3],     # cos == 0: angle = -0.0 -> horizontal
        [-1.0, 0.0, 0.3],
        [0.6, 0.8, 9.0],     # misses the image -> the reference's except branch ([0,0,1,1])
        [0.6, -0.8, 0.0],    # through the centre
        [0.70710678, 0.70710678, 0.0],
        [0.0, 1.0, 5.0],     # vertical line outside the image
    ])
    ao_c = np.concatenate([np.stack([rng.uniform(40, 600, m), rng.uniform(40, 440, m)], 1),
                           np.array([[320, 240], [320, 240], [320, 240], [320, 240], [320, 240], [100, 400], [0, 0], [320, 240]], dtype=np.float64)])
    ao = np.concatenate([ao, ao_special]).astype(np.float32)
    ao_c = ao_c.astype(np.float32)
    bp = [(240.0, 320.0, -np.pi / 2), (240.0, 320.0, 0.0), (0.0, 0.0, np.pi / 4), (479.0, 0.0, -np.pi / 4), (100.5, 200.25, 0.3),
          (100.5, 200.25, -1.2), (-30.0, 700.0, 0.7), (240.0, 320.0, 1e-3), (240.0, 320.0, 1.5), (479.0, 639.0, 0.6435011087932844),
          (1000.0, 1000.0, 0.1), (0.0, 639.0, np.arctan(479.0 / 639.0) - np.pi)]
    bp += [(float(rng.uniform(0, 480)), float(rng.uniform(0, 640)), float(rng.uniform(-1.55, 1.55))) for _ in range(20)]
    return axes, centers, ao, ao_c, bp


def pcd_cases():
    """Inputs of the get_pcd / project2D fixture (vis.py:62-102): mask pixels (x, y) incl. the image corners and the principal point, and
    (unit normal, offset) planes: fronto-parallel, slanted, steep."""
    import numpy as np

    rng = np.random.default_rng(7)
    verts = np.concatenate([np.array([[0, 0], [639, 0], [0, 479], [639, 479], [320, 240], [319, 239]]),
                            np.stack([rng.integers(0, 640, 250), rng.integers(0, 480, 250)], 1)]).astype(np.int64)
    normals = np.array([[0, 0, 1], [0.3, -0.2, 0.93], [-0.6, 0.1, 0.79], [0.05, 0.9, 0.43], [0.7, 0.7, 0.14]], dtype=np.float64)
    normals = (normals / np.linalg.norm(normals, axis=1, keepdims=True)).astype(np.float32)
    offsets = np.array([2.0, 1.3, 3.7, 0.8, 2.2], dtype=np.float32)
    return verts, [(normals[i], offsets[i]) for i in range(len(offsets))]


import numpy as _np

PCD_SHIFT = _np.array([0.013, -0.007, 0.02], dtype=_np.float32)  # fp32 translation applied to the lifted cloud in the pcd_project fixture
