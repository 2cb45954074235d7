"""Builds articulation3d_amd/liba3d_hip.so (the C-ABI kernel library, include/a3d.h) with hipcc for gfx950.

In-tree on purpose: the built .so is git-ignored but travels with the repository snapshot to the GPU
box.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(CSRC, "_obj")
LIB_PATH = os.path.join(PKG_DIR, "liba3d_hip.so")
ARCH = "gfx950"

# file -> extra flags.  The box / paste arithmetic must round like the reference's separate
# mul/add/div operators, hence -ffp-contract=off for those translation units.
SOURCES = {
    "conv_gemm.hip": [],
    "conv_gemm_v2.hip": [],
    "conv_wino.hip": [],
    "conv_wino_fused.hip": [],
    "conv_pw.hip": [],
    "conv_bf16.hip": [],
    "conv_bf16w.hip": [],
    "conv_bf16xs.hip": [],
    "conv_bf16x3.hip": [],
    "conv_bf16x3_wide.hip": [],
    "conv_xs_h2.hip": [],
    "conv_xs_b2b.hip": [],
    "conv_sg_h2.hip": [],
    "conv_ph4p.hip": [],
    "conv_wgrad_tr.hip": [],
    "conv_stem_pool.hip": [],
    "spatial_ops.hip": [],
    "roi_align.hip": ["-ffp-contract=off"],  # sample coordinates ~1e2 px: an FMA-rounded coordinate moves the bilinear weights by 1e-5
    "proposals.hip": ["-ffp-contract=off"],
    "heads_post.hip": ["-ffp-contract=off"],
    "pack.hip": [],
    "conv_wgrad.hip": [],
    "train_sample.hip": [],
    "opt_sweep.hip": ["-ffp-contract=off"],  # the projected pixel index is a truncation: round like the reference's mul / add
    "train_ops.hip": ["-ffp-contract=off"],  # matcher IoU / box deltas round like the reference's separate mul, add, div
}
COMMON = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
COMMON += os.environ.get("A3D_HIPCC_FLAGS", "").split()  # developer builds only (e.g. -DA3D_ABLATIONS: timing-only kernel variants)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stamp(path: str, flags) -> str:
    h = hashlib.sha1()
    for p in (path, os.path.join(CSRC, "a3d_common.h"), os.path.join(CSRC, "conv_common.h"), os.path.join(PKG_DIR, "..", "include", "a3d.h")):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(COMMON + list(flags)).encode())
    return h.hexdigest()


def _compile(src: str, flags) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    stamp_file = obj + ".stamp"
    stamp = _stamp(path, flags)
    if os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj
    cmd = [_hipcc(), *COMMON, *flags, "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile + link (incremental).  Safe to call from several processes at once (one node, shared tree)."""
    import fcntl

    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> str:
    srcs = {k: v for k, v in SOURCES.items() if os.path.exists(os.path.join(CSRC, k))}
    if force:
        for k in srcs:
            s = os.path.join(OBJ_DIR, k.replace(".hip", ".o.stamp"))
            if os.path.exists(s):
                os.remove(s)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda kv: _compile(*kv), srcs.items()))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < newest:
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB_PATH]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB_PATH}")
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
