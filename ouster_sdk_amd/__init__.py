"""ouster_sdk_amd -- MI355X (gfx950) implementation of the Ouster SDK per-pixel hot path.

The product is native: `lib/libouster_hip.so` (HIP kernels behind the C ABI of
include/ouster_hip.h) and `lib/libouster_core_amd.so` (C++ host API with the reference's
names, include/ouster/core/*.h).  The Python modules here are plumbing for tests and the
benchmark: `_capi` (ctypes) and `device` (torch-owned HBM buffers + streams).
"""
from . import _capi  # noqa: F401

__all__ = ["_capi"]
