"""articulation3d_amd: MI355X-native (gfx950) implementation of Articulation3D's per-frame PlaneRCNN
detection path, behind the reference's registry / config / Instances API.

Compute runs in hand-written HIP kernels (articulation3d_amd/csrc, C ABI in include/a3d.h) loaded with
ctypes; there is no CPU or eager-PyTorch fallback.
"""
__version__ = "0.2.0"

import os as _os

# Stream placement as a property of the package.  ROCm maps a process's HIP streams onto GPU_MAX_HW_QUEUES hardware
# queues -- 4 by default --, in the order of their first use; with the default, the package's three side streams plus the caller's and
# RCCL's own streams share queues, and WHICH ones share decides whether two streams overlap or stall each other (the 16-image training
# step: 15.1 ms with the side streams first, 18.1 ms behind three foreign streams).  With 8 queues every stream the package, a host
# application of ordinary size and RCCL use has a queue of its own and the order stops mattering (measured, MI355X, 0 / 3 / 6 foreign
# streams first: 15.0 / 15.1 / 15.1 ms; profiles/r06_stream_queues.txt).  The HIP runtime reads the variable when it initialises, i.e. at
# the first HIP call of the process (importing torch does not make one): importing this package before touching the GPU is enough.  A
# caller's own setting wins; a process whose runtime is already up keeps what it had (articulation3d_amd.streams.queue_setting tells).
_hip_up_at_import = False
try:  # (torch.cuda.is_initialized() is a flag read: it does not touch the GPU)
    import torch as _torch

    _hip_up_at_import = bool(_torch.cuda.is_initialized()) and "GPU_MAX_HW_QUEUES" not in _os.environ
except Exception:  # pragma: no cover
    pass
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import torch_ops  # noqa: E402,F401  registers torch.ops.a3d.* (schemas only: the kernel library is loaded at the first call)
