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
