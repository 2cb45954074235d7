"""COCO run-length codec (column-major runs, compressed ASCII counts) used by `PlaneRCNN_Branch.process` /
`create_instances` to keep the reference's JSON-style record format (pkg/utils/arti_vis.py:66-67,135,179-186;
SURVEY.md A.11).  pycocotools is not available; this restates the published cocoapi encoding."""
from __future__ import annotations

import numpy as np


def encode(mask: np.ndarray) -> dict:
    h, w = mask.shape
    flat = np.asarray(mask, dtype=np.uint8).reshape(-1, order="F")
    if flat.size == 0:
        return {"size": [h, w], "counts": ""}
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    bounds = np.concatenate(([0], change, [flat.size]))
    runs = np.diff(bounds).tolist()
    if flat[0] == 1:
        runs = [0] + runs
    return {"size": [h, w], "counts": _to_string(runs)}


class RLEDict(dict):
    """An RLE dict (`{"size": [h, w], "counts": "..."}`: serialises and compares like one) that remembers the DEVICE mask it
    encodes, so that a consumer which needs the dense mask again -- `create_instances` (pkg/utils/arti_vis.py:179-186) -- can
    take it from the device instead of decoding the string on the host.  Only a dict that still IS this object carries it."""
    __slots__ = ("_dense",)


def launch_encode_device(masks):
    """First half of encode_device: launches a3d_mask_rle and returns (int32 device buffer, D, h, w, cap, uint8 masks); the caller
    copies the buffer to the host (alone or packed with other results) and hands it to finish_encode_device."""
    import torch

    from .. import _lib

    D, h, w = masks.shape
    m = masks.contiguous()
    m = m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
    cap = 2048
    buf = torch.empty((2 * D + D * cap,), device=m.device, dtype=torch.int32)
    if D:
        _lib.check(_lib.lib().a3d_mask_rle(m.data_ptr(), D, h, w, cap, buf[2 * D:].data_ptr(), buf.data_ptr(), buf[D:].data_ptr(),
                                           torch.cuda.current_stream().cuda_stream), "a3d_mask_rle")
    return buf, D, h, w, cap, m


def finish_encode_device(host: np.ndarray, D: int, h: int, w: int, cap: int, m, keep_dense: bool = False) -> list:
    """host: the int32 buffer of launch_encode_device on the host.  Masks with more than `cap` run boundaries are re-encoded
    through encode_device with a larger buffer."""
    if D == 0:
        return []
    cf_h = host[:2 * D].reshape(2, D)
    if int(cf_h[0].max()) > cap:
        return encode_device(m, keep_dense=keep_dense, cap=int(cf_h[0].max()))
    pos_h = host[2 * D:].reshape(D, cap)
    runs_list = []
    for d in range(D):
        n = int(cf_h[0, d])
        bounds = np.concatenate(([0], pos_h[d, :n].astype(np.int64), [h * w]))
        runs = np.diff(bounds)
        runs_list.append(np.concatenate(([0], runs)) if cf_h[1, d] else runs)
    out = []
    for d, counts in enumerate(_to_strings_batch(runs_list)):
        r = RLEDict(size=[h, w], counts=counts)
        r._dense = m[d] if keep_dense else None
        out.append(r)
    return out


def encode_device(masks, keep_dense: bool = False, cap: int = 2048) -> list:
    """masks: CUDA tensor [D,H,W] (bool / uint8, non-zero = set) -> the list of `encode(mask)` dicts, identical strings.
    The run boundaries are found on the device (a3d_mask_rle: column-major scan of each mask by one workgroup), so what crosses
    PCIe is a few hundred integers per mask instead of the 307 KB mask, and the host builds the count strings from them."""
    import torch

    from .. import _lib
    from ..structures import to_host

    D, h, w = masks.shape
    if D == 0:
        return []
    m = masks.contiguous()
    m = m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
    while True:  # ONE buffer [count D | first D | positions D x cap] and ONE device -> host copy of it
        buf = torch.empty((2 * D + D * cap,), device=m.device, dtype=torch.int32)
        _lib.check(_lib.lib().a3d_mask_rle(m.data_ptr(), D, h, w, cap, buf[2 * D:].data_ptr(), buf.data_ptr(), buf[D:].data_ptr(),
                                           torch.cuda.current_stream().cuda_stream), "a3d_mask_rle")
        host = to_host(buf).numpy()
        mx = int(host[:D].max())
        if mx <= cap:
            break
        cap = mx  # (a mask with more run boundaries than that: once more with room for all of them)
    return finish_encode_device(host, D, h, w, cap, m, keep_dense)


def decode(rle: dict) -> np.ndarray:
    h, w = rle["size"]
    counts = rle["counts"]
    runs = _from_string(counts) if isinstance(counts, (str, bytes)) else list(counts)
    vals = np.zeros(len(runs), dtype=np.uint8)
    vals[1::2] = 1
    flat = np.repeat(vals, runs)
    assert flat.size == h * w, (flat.size, h, w)
    return flat.reshape((h, w), order="F")


def _to_string(cnts) -> str:
    out = []
    for i, x in enumerate(cnts):
        x = int(x)
        if i > 2:
            x -= int(cnts[i - 2])
        more = True
        while more:
            c = x & 0x1F
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(chr(c + 48))
    return "".join(out)


def _to_strings_batch(runs_list) -> list:
    """`_to_string` for several run-length lists at once, vectorised (cocoapi rleToString: every count -- from the fourth on as
    the difference to the count two places earlier -- in 5-bit groups, little end first, bit 5 = "more", sign-extended)."""
    if not runs_list:
        return []
    xs, lens = [], []
    for r in runs_list:
        r = np.asarray(r, dtype=np.int64)
        x = r.copy()
        if len(r) > 3:
            x[3:] -= r[1:-2]
        xs.append(x)
        lens.append(len(r))
    x = np.concatenate(xs)
    n = len(x)
    cols = []
    alive = np.ones(n, dtype=bool)
    while alive.any():
        c = x & 0x1F
        x = x >> 5
        more = np.where((c & 0x10) != 0, x != -1, x != 0) & alive
        cols.append(np.where(alive, (c | np.where(more, 0x20, 0)) + 48, 0).astype(np.uint8))
        alive = more
    table = np.stack(cols, 1)                      # [n, rounds]: the characters of every count, 0 = none
    nchar = (table != 0).sum(1)
    flat = table.reshape(-1)
    flat = flat[flat != 0].tobytes().decode("ascii")
    ends = np.cumsum(nchar)
    out, o, e0 = [], 0, 0
    for ln in lens:
        e1 = int(ends[o + ln - 1]) if ln else e0
        out.append(flat[e0:e1])
        e0, o = e1, o + ln
    return out


def _from_string(s) -> list:
    if isinstance(s, bytes):
        s = s.decode("ascii")
    cnts = []
    p = 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return cnts
