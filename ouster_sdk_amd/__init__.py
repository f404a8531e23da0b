"""ouster_sdk_amd -- MI355X (gfx950) implementation of the Ouster SDK per-pixel hot path.

The product is native: `lib/libouster_hip.so` (HIP kernels behind the C ABI of
include/ouster_hip.h) and `lib/libouster_core_amd.so` (C++ host API with the reference's
names, include/ouster/core/*.h).  Python pieces:
  core      pybind11 module over the C++ mirror (XYZLut, destagger, FrameBatcher, LidarFrame,
            PacketFormat ...) -- the `ouster.sdk.core` call shapes for this path
  sdk       device-aware front of `core`: numpy in -> numpy out, CUDA tensor in -> CUDA tensor out
  _capi     ctypes binding of the C ABI
  device    torch-owned HBM buffers + streams around the C ABI (batched, device resident)
  parallel  frame sharding across GPUs (torch.distributed)
"""
try:  # torch first: its bundled HIP runtime must be the one libouster_hip.so binds to
    import torch as _torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for the pure C++/ctypes users
    _torch = None

from . import _capi  # noqa: F401,E402

__all__ = ["_capi"]
