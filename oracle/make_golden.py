#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE's own modules in the build container.

Run here (needs /root/reference; it never travels to the GPU box -- only the small .npz vectors do):
    python oracle/make_golden.py

What can be imported from the reference:
  * articulation3d/layers/mask_ops.py            -- imports only numpy / torch / PIL: used as is;
  * modeling/roi_heads/plane_head.py, axis_head.py, modeling/depth_net/depth_head.py
        -- import seven detectron2 / fvcore symbols at module scope (Conv2d, ShapeSpec, cat, get_norm,
        Registry, weight_init, smooth_l1_loss).  detectron2 / fvcore are not installed, so those seven names
        are provided by the minimal definitions below (a Conv2d that is nn.Conv2d + optional norm + activation,
        exactly the documented detectron2 wrapper; the two fvcore initialisers).  The arithmetic exercised --
        the layer lists, forward order, flatten order, F.normalize, cat, BatchNorm eps, upsampling and the two
        bilinear resizes -- is all the reference's own code.
  * data/planercnn_transforms.py (axis_to_angle_offset, angle_offset_to_axis, get_boundary_point) and utils/vis.py (get_pcd,
        project2D) -- pure numpy / torch functions in modules whose OTHER code imports detectron2, cv2, pytorch3d, mapbox_earcut,
        imageio, pycocotools and two sibling modules at module scope.  Those imports are satisfied by NAME-ONLY placeholders
        (_install_placeholder_names: empty modules whose attributes are inert objects; no arithmetic is substituted -- the five
        functions never touch them).  They are the helpers of the temporal optimiser (SURVEY 8c fixture 4, 8f-3).
Everything else on the path (backbone, RPN, ROIAlign, NMS, box / mask heads) has no source under
/root/reference: parity unpinned (oracle/planercnn_oracle.py header).

Fixtures hold inputs' checksums + expected outputs only; weights and inputs are regenerated from seeds by
oracle.golden_inputs (committed), so the files stay small.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/articulation3d"
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


def _install_d2_names():
    """The seven third-party names the reference heads import at module scope."""

    class ShapeSpec:
        def __init__(self, channels=None, height=None, width=None, stride=None):
            self.channels, self.height, self.width, self.stride = channels, height, width, stride

    class Conv2d(nn.Conv2d):  # detectron2.layers.Conv2d: conv -> norm -> activation
        def __init__(self, *args, **kwargs):
            norm = kwargs.pop("norm", None)
            activation = kwargs.pop("activation", None)
            super().__init__(*args, **kwargs)
            self.norm, self.activation = norm, activation

        def forward(self, x):
            x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
            if self.norm is not None:
                x = self.norm(x)
            if self.activation is not None:
                x = self.activation(x)
            return x

    def get_norm(norm, out_channels):
        assert not norm, "reference configs use NORM ''"
        return None

    class Registry:
        def __init__(self, name):
            self._name, self._map = name, {}

        def register(self, obj=None):
            def deco(o):
                self._map[o.__name__] = o
                return o

            return deco if obj is None else deco(obj)

        def get(self, name):
            return self._map[name]

    wi = types.ModuleType("fvcore.nn.weight_init")

    def c2_msra_fill(m):
        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)

    def c2_xavier_fill(m):
        nn.init.kaiming_uniform_(m.weight, a=1)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)

    wi.c2_msra_fill, wi.c2_xavier_fill = c2_msra_fill, c2_xavier_fill
    mods = {
        "detectron2": types.ModuleType("detectron2"),
        "detectron2.layers": types.ModuleType("detectron2.layers"),
        "detectron2.utils": types.ModuleType("detectron2.utils"),
        "detectron2.utils.registry": types.ModuleType("detectron2.utils.registry"),
        "fvcore": types.ModuleType("fvcore"),
        "fvcore.nn": types.ModuleType("fvcore.nn"),
        "fvcore.nn.weight_init": wi,
    }
    mods["detectron2.layers"].Conv2d = Conv2d
    mods["detectron2.layers"].ShapeSpec = ShapeSpec
    mods["detectron2.layers"].cat = lambda ts, dim=0: torch.cat(ts, dim)
    mods["detectron2.layers"].get_norm = get_norm
    mods["detectron2.utils.registry"].Registry = Registry
    mods["fvcore.nn"].weight_init = wi
    mods["fvcore.nn"].smooth_l1_loss = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError)
    sys.modules.update(mods)
    return ShapeSpec


class _Inert:
    """A name that resolves and does nothing: any attribute is another inert name, calling it raises."""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Inert()

    def __call__(self, *a, **k):
        raise NotImplementedError("placeholder for a third-party name the golden functions never use")


def _install_placeholder_names():
    """Module-scope imports of data/planercnn_transforms.py and utils/vis.py that the five pinned functions never use."""

    class _Mod(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return _Inert()

    for name in ("detectron2.data", "detectron2.data.detection_utils", "detectron2.data.transforms", "detectron2.structures",
                 "detectron2.structures.masks", "cv2", "mapbox_earcut", "imageio", "pytorch3d", "pytorch3d.structures",
                 "pytorch3d.renderer", "pytorch3d.renderer.mesh", "pycocotools", "pycocotools.mask",
                 "refpkg", "refpkg.utils", "refpkg.utils.mesh_utils", "refpkg.utils.pycococreatortools"):
        if name not in sys.modules:
            m = _Mod(name)
            m.__path__ = []  # (a package: `from x.y import z` resolves through sys.modules)
            sys.modules[name] = m
    if not hasattr(sys.modules["detectron2"], "data"):
        sys.modules["detectron2"].data = sys.modules["detectron2.data"]
        sys.modules["detectron2"].structures = sys.modules["detectron2.structures"]


def _load(path, name):
    import importlib.util

    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def _head_cfg():
    h = _Cfg(NUM_CONV=4, CONV_DIM=256, NUM_FC=1, FC_DIM=1024, PARAM_DIM=3, NORM="", NORMAL_ONLY=True, LOSS_WEIGHT=1.0,
             SMOOTH_L1_BETA=0.0)
    return _Cfg(MODEL=_Cfg(ROI_PLANE_HEAD=h, ROI_AXIS_HEAD=h, DEPTH_HEAD=_Cfg(LOSS_WEIGHT=1.0), FREEZE=[]))


class _FakeInstances(list):
    """len()-able holder the reference's *_inference functions set attributes on."""


class _Inst:
    def __init__(self, n):
        self._n = n

    def __len__(self):
        return self._n


def main():
    from oracle import golden_inputs as G

    os.makedirs(OUT, exist_ok=True)
    ShapeSpec = _install_d2_names()
    torch.set_num_threads(8)

    # ---- (1) mask paste: reference module used as is -------------------------------------------------
    mo = _load(os.path.join(REF, "articulation3d", "layers", "mask_ops.py"), "ref_mask_ops")
    masks, boxes, hw = G.paste_case()
    with torch.no_grad():
        out = mo.paste_masks_in_image(masks, boxes, hw, threshold=0.5)
    np.savez_compressed(os.path.join(OUT, "paste_masks.npz"), masks_sum=float(masks.double().sum()), boxes=boxes.numpy(),
                        image_hw=np.array(hw), packed=np.packbits(out.numpy().astype(np.uint8)), shape=np.array(out.shape))
    print("paste_masks", tuple(out.shape), int(out.sum()))

    # ---- (2) plane head ------------------------------------------------------------------------------
    ph = _load(os.path.join(REF, "articulation3d", "modeling", "roi_heads", "plane_head.py"), "ref_plane_head")
    cfg = _head_cfg()
    P = G.head_params("plane")
    m = ph.PlaneRCNNConvFCHead(cfg, ShapeSpec(channels=256, height=14, width=14)).eval()
    m.load_state_dict({k.replace("roi_heads.plane_head.", ""): v for k, v in P.items()})
    x = G.head_input()
    inst = [_Inst(2), _Inst(3)]
    with torch.no_grad():
        m(x, inst)
    pred = torch.cat([i.pred_plane for i in inst]).numpy()
    inst = [_Inst(2), _Inst(3)]
    with torch.no_grad():
        m.double()(x.double(), inst)  # the reference module evaluated in float64: ground truth for the fp32 error budget
    pred64 = torch.cat([i.pred_plane for i in inst]).numpy()
    np.savez_compressed(os.path.join(OUT, "plane_head.npz"), x_sum=float(x.double().sum()), pred_plane=pred, pred_plane_f64=pred64)
    print("plane_head", pred.shape, pred[0])

    # ---- (3) axis head -------------------------------------------------------------------------------
    import io
    import contextlib

    ah = _load(os.path.join(REF, "articulation3d", "modeling", "roi_heads", "axis_head.py"), "ref_axis_head")
    P = G.head_params("axis")
    with contextlib.redirect_stdout(io.StringIO()):  # the reference prints its smooth-l1 beta
        m = ah.PlaneRCNNConvFCHead(cfg, ShapeSpec(channels=256, height=14, width=14)).eval()
    m.load_state_dict({k.replace("roi_heads.axis_head.", ""): v for k, v in P.items()})
    inst = [_Inst(2), _Inst(3)]
    with torch.no_grad():
        m(x, inst)
    rot = torch.cat([i.pred_rot_axis for i in inst]).numpy()
    tran = torch.cat([i.pred_tran_axis for i in inst]).numpy()
    inst = [_Inst(2), _Inst(3)]
    with torch.no_grad():
        m.double()(x.double(), inst)
    rot64 = torch.cat([i.pred_rot_axis for i in inst]).numpy()
    tran64 = torch.cat([i.pred_tran_axis for i in inst]).numpy()
    np.savez_compressed(os.path.join(OUT, "axis_head.npz"), x_sum=float(x.double().sum()), pred_rot_axis=rot, pred_tran_axis=tran,
                        pred_rot_axis_f64=rot64, pred_tran_axis_f64=tran64)
    print("axis_head", rot.shape, tran.shape)

    # ---- (4) depth head ------------------------------------------------------------------------------
    dh = _load(os.path.join(REF, "articulation3d", "modeling", "depth_net", "depth_head.py"), "ref_depth_head")
    P = G.head_params("depth")
    m = dh.PlaneRCNNDepthHead(cfg).eval()
    missing, unexpected = m.load_state_dict({k.replace("depth_head.", ""): v for k, v in P.items()}, strict=False)
    assert not unexpected and all("num_batches_tracked" in k for k in missing), (missing, unexpected)
    feats = G.depth_features()
    with torch.no_grad():
        d = m(feats)
    with torch.no_grad():
        d64 = m.double()({k: v.double() for k, v in feats.items()})
    np.savez_compressed(os.path.join(OUT, "depth_head.npz"), feats_sum=float(sum(v.double().sum() for v in feats.values())),
                        depth_strided=d[:, ::16, ::16].numpy(), depth_strided_f64=d64[:, ::16, ::16].numpy(), depth_sum=float(d.double().sum()),
                        depth_abs_sum=float(d.double().abs().sum()), shape=np.array(d.shape))
    print("depth_head", tuple(d.shape), float(d.mean()))

    # ---- (5) axis <-> (angle, offset) transforms and (6) point-cloud lift / projection: the temporal optimiser's helpers --------
    _install_placeholder_names()
    tr = _load(os.path.join(REF, "articulation3d", "data", "planercnn_transforms.py"), "ref_planercnn_transforms")
    axes, centers, ao, ao_centers, bp = G.axis_cases()
    fwd = tr.axis_to_angle_offset([list(map(float, a)) for a in axes], torch.tensor(centers), mine=False).numpy()
    fwd_mine = tr.axis_to_angle_offset([list(map(float, a)) for a in axes], torch.tensor(centers), mine=True).numpy()
    with_none = tr.axis_to_angle_offset([None, list(map(float, axes[0]))], torch.tensor(centers[:2])).numpy()
    back = tr.angle_offset_to_axis(torch.tensor(ao), torch.tensor(ao_centers)).numpy()
    # round trip through the reference's own two functions (what the optimiser does with a predicted axis: opt_utils.py:396-400)
    back_rt = tr.angle_offset_to_axis(torch.tensor(fwd[:, :3]), torch.tensor(centers)).numpy()
    pts = []
    for y, x, ang in bp:
        p1, p2 = tr.get_boundary_point(y, x, ang, 480, 640)
        pts.append([-1] * 4 if p1 is None else [p1[0], p1[1], p2[0], p2[1]])
    np.savez_compressed(os.path.join(OUT, "axis_transforms.npz"), axes=axes, centers=centers, angle_offset=fwd, angle_offset_mine=fwd_mine,
                        with_none=with_none, ao=ao, ao_centers=ao_centers, axis_back=back, axis_round_trip=back_rt,
                        boundary_in=np.asarray(bp, dtype=np.float64), boundary_out=np.asarray(pts, dtype=np.float64))
    print("axis_transforms", fwd.shape, back.shape, len(pts))

    sys.modules["refpkg.utils"].__path__ = [os.path.join(REF, "articulation3d", "utils")]
    import importlib.util
    spec = importlib.util.spec_from_file_location("refpkg.utils.vis", os.path.join(REF, "articulation3d", "utils", "vis.py"))
    vis = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(vis)
    verts, planes = G.pcd_cases()
    pcds, projs, projs32, projs_sh = [], [], [], []
    shift = G.PCD_SHIFT  # a translation hypothesis: the lifted points leave the pixel edges the identity puts them on
    for normal, offset in planes:
        pcd = vis.get_pcd(verts, normal, offset)                      # float64 (numpy)
        pcds.append(pcd)
        projs.append(vis.project2D(pcd))                              # numpy branch on the float64 cloud
        projs32.append(vis.project2D(pcd.astype(np.float32)))         # numpy branch on the fp32 cloud the optimiser keeps (K stays float64)
        projs_sh.append(vis.project2D((pcd.astype(np.float32) + shift).astype(np.float32)))
    np.savez_compressed(os.path.join(OUT, "pcd_project.npz"), verts=verts, normals=np.stack([p[0] for p in planes]),
                        offsets=np.array([p[1] for p in planes]), pcd=np.stack(pcds), proj=np.stack(projs), proj_from_f32=np.stack(projs32),
                        shift=shift, proj_shifted=np.stack(projs_sh))
    print("pcd_project", np.stack(pcds).shape)


if __name__ == "__main__":
    main()
