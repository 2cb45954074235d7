"""articulation3d_amd: MI355X-native (gfx950) implementation of Articulation3D's per-frame PlaneRCNN
detection path, behind the reference's registry / config / Instances API.

Compute runs in hand-written HIP kernels (articulation3d_amd/csrc, C ABI in include/a3d.h) loaded with
ctypes; there is no CPU or eager-PyTorch fallback.
"""
__version__ = "0.2.0"

import os as _os

# Stream placement (measured in round 6, profiles/r06_stream_queues.txt, r06_queue_matrix.txt).  ROCm maps a process's HIP
# streams onto GPU_MAX_HW_QUEUES hardware queues -- 4 by default -- in the order of their first use, and which streams share a queue decides
# whether they overlap or stall each other (the 16-image training step: 15.1 ms with the package's pool first, 18.1 ms behind three foreign
# streams).  Raising the queue count does make that order-independent (6 or 8 queues: 15.0 / 15.1 / 15.1 ms with 0 / 3 / 6 foreign streams
# first) -- but with 6 or more queues the cross-stream waits of the training step's gradient exchange (weight-gradient stream ->
# communication stream -> RCCL's stream -> back) cost 5-6 ms per 5.4 ms step instead of 0.36 (one-rank RCCL group; 5 queues: 0.24 ms, but
# then ONE first-use order is slow again: 17.96 ms).  No queue count is good for both, so the package does NOT set the variable: the
# runtime's default stays, and the rule stays that the package's pool is used first (articulation3d_amd.streams.side; bench.py,
# tools/train_bench.py and tools/inference.py call it before their own streams and before RCCL initialises).
_hip_up_at_import = False
try:  # (torch.cuda.is_initialized() is a flag read: it does not touch the GPU)
    import torch as _torch

    _hip_up_at_import = bool(_torch.cuda.is_initialized())
except Exception:  # pragma: no cover
    pass

from . import torch_ops  # noqa: E402,F401  registers torch.ops.a3d.* (schemas only: the kernel library is loaded at the first call)
