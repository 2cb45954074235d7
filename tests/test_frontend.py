"""Input front end: cv2.resize(im, (640, 480)) + BGR flip + normalise (tools/inference.py:216-218).
CPU part: the oracle's restatement of OpenCV's published fixed-point bilinear (parity unpinned: cv2 is absent here) checked
through properties; GPU part: `a3d_preprocess_resize_u8` bit-exact against it, edge sizes included."""
import numpy as np
import pytest
import torch

from oracle.resize_oracle import cv2_resize_linear_u8, frontend

MEAN, STD = (103.53, 116.28, 123.675), (1.0, 1.0, 1.0)
SIZES = [(480, 640), (960, 1280), (720, 1280), (1080, 1920), (97, 131), (240, 320), (481, 639), (1, 1), (2, 3)]


def _img(h, w, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def _float_bilinear(img, Wd, Hd):
    Hs, Ws = img.shape[:2]
    fx, fy = (np.arange(Wd) + 0.5) * Ws / Wd - 0.5, (np.arange(Hd) + 0.5) * Hs / Hd - 0.5
    x0, y0 = np.floor(fx).astype(int), np.floor(fy).astype(int)
    wx, wy = np.where((x0 < 0) | (x0 >= Ws - 1), 0, fx - x0), np.where((y0 < 0) | (y0 >= Hs - 1), 0, fy - y0)
    x0c, x1c, y0c, y1c = np.clip(x0, 0, Ws - 1), np.clip(x0 + 1, 0, Ws - 1), np.clip(y0, 0, Hs - 1), np.clip(y0 + 1, 0, Hs - 1)
    I = img.astype(np.float64)
    top = I[y0c][:, x0c] * (1 - wx)[None, :, None] + I[y0c][:, x1c] * wx[None, :, None]
    bot = I[y1c][:, x0c] * (1 - wx)[None, :, None] + I[y1c][:, x1c] * wx[None, :, None]
    return top * (1 - wy)[:, None, None] + bot * wy[:, None, None]


@pytest.mark.parametrize("hw", SIZES)
def test_resize_oracle_properties(hw):
    im = _img(*hw)
    out = cv2_resize_linear_u8(im, (640, 480))
    assert out.shape == (480, 640, 3) and out.dtype == np.uint8
    if hw == (480, 640):
        assert np.array_equal(out, im)  # equal sizes: copy
    elif hw == (960, 1280):  # exact 2x: INTER_AREA fast path = rounded 2x2 mean
        assert np.array_equal(out, np.floor(im.reshape(480, 2, 640, 2, 3).astype(np.float64).mean((1, 3)) + 0.5).astype(np.uint8))
    else:  # fixed-point bilinear stays within one grey level of the real-valued bilinear interpolation
        assert np.abs(out.astype(np.float64) - _float_bilinear(im, 640, 480)).max() <= 1.0
    const = np.full(hw + (3,), 137, dtype=np.uint8)
    assert (cv2_resize_linear_u8(const, (640, 480)) == 137).all()  # partition of unity survives the 11-bit coefficients
    assert out.min() >= im.min() and out.max() <= im.max()


def test_frontend_oracle_layout():
    x, res = frontend(_img(97, 131)[None], MEAN, STD)
    assert x.shape == (1, 480, 640, 3) and res.shape == (1, 480, 640, 3)
    assert np.array_equal(x[0, ..., 0] + np.float32(MEAN[0]), res[0, ..., 2].astype(np.float32))  # output channel 0 = B = source channel 2


@pytest.mark.gpu
@pytest.mark.parametrize("hw", SIZES)
def test_hip_resize_frontend_bit_exact(hw):
    from articulation3d_amd import ops

    frames = np.stack([_img(*hw, seed=s) for s in (1, 2)])
    x4, u8 = ops.preprocess_resize_u8(torch.from_numpy(frames).cuda(), MEAN, STD, (480, 640), swap_rb=True, want_u8=True)
    want_x, want_u8 = frontend(frames, MEAN, STD)
    assert torch.equal(u8.cpu(), torch.from_numpy(want_u8))  # integer arithmetic: bit-exact
    assert torch.equal(x4[..., :3].cpu(), torch.from_numpy(want_x)) and float(x4[..., 3].abs().sum()) == 0
    # equal size + no swap == the plain normalisation kernel
    if hw == (480, 640):
        a = ops.preprocess_resize_u8(torch.from_numpy(frames).cuda(), MEAN, STD, (480, 640), swap_rb=False)
        assert torch.equal(a, ops.preprocess_u8hwc(torch.from_numpy(frames).cuda(), MEAN, STD))


@pytest.mark.gpu
def test_detect_clip_from_reader_frames_equals_host_side_resize(hip_model, oracle):
    """The fused front end changes nothing downstream: raw RGB 720p frames through detect_clip(source_rgb=True) == the same frames
    resized + flipped on the host (oracle restatement of cv2.resize) through the uint8 BGR entry."""
    from articulation3d_amd.pipeline import detect_clip

    model = hip_model
    model.roi_heads.box_predictor.test_score_thresh = 0.0
    raw = np.stack([_img(720, 1280, seed=s) for s in (5, 6, 7)])
    a = detect_clip(model, raw, batch=2, conf_threshold=0.35, source_rgb=True)
    host = np.ascontiguousarray(np.stack([cv2_resize_linear_u8(f, (640, 480)) for f in raw])[..., ::-1])
    b = detect_clip(model, host, batch=2, conf_threshold=0.35)
    assert len(a) == len(b) == 3
    for p, q in zip(a, b):
        assert len(p) == len(q) > 0 and torch.equal(p.pred_boxes.tensor, q.pred_boxes.tensor) and torch.equal(p.pred_masks, q.pred_masks)
        assert torch.equal(p.pred_planes, q.pred_planes)


@pytest.mark.gpu
def test_device_rle_encode_equals_the_host_codec():
    """a3d_mask_rle + rle.encode_device (the `segmentation` strings of PlaneRCNN_Branch.process, arti_vis.py:66-67) against the host
    restatement of cocoapi's encoder: identical dicts on blobs, empty / full masks, a set first pixel, boundaries at column ends, and
    a mask with more run boundaries than the first buffer holds (checkerboard: one per pixel)."""
    import numpy as np
    import torch

    from articulation3d_amd.utils import rle

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    rng = np.random.default_rng(5)
    H, W = 480, 640
    masks = np.zeros((9, H, W), dtype=np.uint8)
    yy, xx = np.mgrid[:H, :W]
    for d in range(4):  # blobs
        cy, cx, ry, rx = rng.uniform(50, 430), rng.uniform(50, 590), rng.uniform(10, 200), rng.uniform(10, 250)
        masks[d] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1).astype(np.uint8)
    masks[4] = 1                      # full
    masks[6, 0, 0] = 1                # only the first pixel (leading zero-length run of zeros)
    masks[6, H - 1, 0] = 1            # a run that ends exactly at a column end
    masks[6, 0, 1] = 1                # ... and one that starts at the next column's top: they merge in column-major order
    masks[7, :, W - 1] = 1            # last column
    masks[8] = ((yy + xx) & 1).astype(np.uint8)  # checkerboard: H*W - 1 boundaries
    got = rle.encode_device(torch.from_numpy(masks).cuda())
    for d in range(9):
        want = rle.encode(masks[d])
        assert got[d] == want, d
        assert (rle.decode(got[d]) == masks[d]).all()
    got_bool = rle.encode_device(torch.from_numpy(masks[:4].astype(bool)).cuda())
    assert got_bool == got[:4]
    assert rle.encode_device(torch.zeros((0, H, W), dtype=torch.uint8, device="cuda")) == []


def test_every_rank_of_an_image_directory_reads_its_block_even_an_empty_one(tmp_path):
    """tools/inference.py's reader under torchrun: shard_range uses ceil blocks, so with 9 frames on 8 ranks the blocks are
    2,2,2,2,1,0,0,0 -- the trailing ranks' EMPTY blocks come back as (0, H, W, 3) arrays at the clip's size (a rank that raised
    there died before the gather and took the job down)."""
    import importlib.util
    import os

    from PIL import Image

    from articulation3d_amd.parallel import shard_range

    spec = importlib.util.spec_from_file_location("a3d_inference_tool", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "inference.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    for i in range(9):
        Image.fromarray(_img(24, 32, seed=i)).save(tmp_path / f"f{i:03d}.png")
    assert tool.count_frames(str(tmp_path)) == 9
    whole = tool.read_frames(str(tmp_path))
    assert whole.shape == (9, 24, 32, 3)
    got = []
    for rank in range(8):
        lo, hi = shard_range(9, rank, 8)
        blk = tool.read_frames(str(tmp_path), lo, hi)
        assert blk.dtype == np.uint8 and blk.shape == (hi - lo, 24, 32, 3), (rank, blk.shape)
        got.append(blk)
    assert sum(len(b) == 0 for b in got) == 3
    assert np.array_equal(np.concatenate(got), whole)
