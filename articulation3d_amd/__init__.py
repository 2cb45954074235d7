"""articulation3d_amd: MI355X-native (gfx950) implementation of Articulation3D's per-frame PlaneRCNN
detection path, behind the reference's registry / config / Instances API.

Compute runs in hand-written HIP kernels (articulation3d_amd/csrc, C ABI in include/a3d.h) loaded with
ctypes; there is no CPU or eager-PyTorch fallback.
"""
__version__ = "0.2.0"

from . import torch_ops  # noqa: E402,F401  registers torch.ops.a3d.* (schemas only: the kernel library is loaded at the first call)
