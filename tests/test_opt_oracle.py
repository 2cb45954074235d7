"""CPU suite: the optimiser oracle (oracle/opt_oracle.py) against independent statements (scipy rotations, analytic
geometry) and the host-side helpers of the product's optimiser against it."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation


@pytest.fixture(scope="module")
def OO():
    from oracle import opt_oracle

    return opt_oracle


def test_axis_angle_to_matrix_is_the_rotation_matrix(OO):
    rng = np.random.default_rng(0)
    aa = rng.normal(size=(20, 3))
    aa[0] *= 1e-9  # small-angle branch
    assert np.allclose(OO.axis_angle_to_matrix(aa), Rotation.from_rotvec(aa).as_matrix(), atol=1e-12)
    # pytorch3d's Rotate multiplies row vectors: the oracle hands out the transpose, i.e. a rotation by -angle
    hyps, pivot = OO.rotation_hypotheses(np.array([0.3], np.float32), np.array([0.0, 0.0, 1.0]), np.zeros(3))
    assert np.allclose(hyps[0][0], Rotation.from_rotvec([0, 0, -0.3]).as_matrix(), atol=1e-6)


def test_get_pcd_points_lie_on_the_plane_and_reproject(OO):
    normal = np.array([0.1, -0.2, 0.97], dtype=np.float32)
    normal /= np.linalg.norm(normal)
    verts = np.array([[10, 20], [320, 240], [600, 400]])
    p = OO.get_pcd(verts, normal, 2.5)
    assert np.allclose(p @ normal.astype(np.float64), 2.5, atol=1e-6)          # n . X = offset
    uv = OO.project2d(p.astype(np.float32))
    assert np.allclose(uv, verts, atol=1e-3)


def test_axis_round_trip_and_product_helpers_agree(OO):
    from articulation3d_amd.utils import opt_utils as PU

    centers = np.array([[300.0, 200.0], [100.0, 400.0], [500.0, 100.0]], dtype=np.float32)
    axes = np.array([[250.0, 0.0, 250.0, 479.0], [0.0, 100.0, 639.0, 300.0], [50.0, 0.0, 400.0, 479.0]])
    ao = OO.axis_to_angle_offset(axes, centers)
    assert np.allclose(PU.axis_to_angle_offset(axes.tolist(), torch.from_numpy(centers)).numpy(), ao, atol=1e-6)
    back = OO.angle_offset_to_axis(ao[:, :3], centers)
    assert np.array_equal(PU.angle_offset_to_axis(torch.from_numpy(ao[:, :3]), torch.from_numpy(centers)).numpy(), back)
    for (x1, y1, x2, y2), (u1, v1, u2, v2) in zip(axes, back):  # same line: both recovered end points lie on it (to a pixel)
        n = np.array([y1 - y2, x2 - x1])
        n = n / np.linalg.norm(n)
        for (u, v) in ((u1, v1), (u2, v2)):
            assert abs(n @ np.array([u - x1, v - y1])) < 1.5
    assert np.allclose(PU._axis_angle_to_matrix(np.array([[0.1, 0.2, 0.3]])), OO.axis_angle_to_matrix(np.array([[0.1, 0.2, 0.3]])))
    assert np.array_equal(PU.ROT_ANGLES.numpy(), OO.ROT_ANGLES) and np.array_equal(PU.TRANS_STEPS.numpy(), OO.TRANS_STEPS)
    assert len(OO.ROT_ANGLES) == 45 and len(OO.ROT_ANGLES_FINAL) == 30 and len(OO.TRANS_STEPS) == 20


def test_projection_sweep_on_a_fronto_parallel_plane(OO):
    """A plane facing the camera at depth 2: translating by s along x moves the mask by f*s/2 pixels."""
    mask = np.zeros((480, 640), bool)
    mask[200:280, 300:380] = True
    normal, offset = np.array([0.0, 0.0, 1.0], np.float32), np.float32(2.0)
    ys, xs = np.nonzero(mask)
    pcd = OO.get_pcd(np.stack([xs, ys], 1), normal, offset).astype(np.float32)
    hyps, pivot = OO.translation_hypotheses(np.array([0.0, 0.2], np.float32), np.array([1.0, 0.0, 0.0]))
    out = OO.project_masks(pcd, hyps, pivot)
    shift = 517.97 * 0.2 / 2.0
    ys1, xs1 = np.nonzero(out[1])
    assert abs((xs1.mean() - xs.mean()) - shift) < 1.0 and abs(ys1.mean() - ys.mean()) < 1.0
    iou = OO.mask_ious(mask.astype(np.float32), out)
    assert iou[0] > 0.7 and iou[1] < iou[0]  # (the identity hypothesis is speckled by the truncation: not exactly 1)
