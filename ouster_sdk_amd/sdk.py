"""Device-aware front of the Python drop-in.

`ouster_sdk_amd.core` (pybind11) mirrors the reference's numpy-in / numpy-out call shapes
(python/src/cpp/client/processing.cpp:340-357 XYZLut, :527-638 destagger).  The wrappers here accept
the same arguments and, when the image is a CUDA torch tensor, keep everything in HBM: the result is
a torch tensor on the same device (exportable through ``__dlpack__``), computed by the same kernels on
torch's current stream, with no PCIe round trip.  numpy inputs go to `core` unchanged.

    lut = sdk.XYZLut(info)                 # like ouster.sdk.core.XYZLut(metadata)
    xyz = lut(range_img)                   # numpy (h, w) -> numpy (h, w, 3); cuda tensor -> cuda tensor
    img = sdk.destagger(info, field)       # numpy or cuda tensor, (h, w) or (h, w, n)
"""
import ctypes as C

import numpy as np

from . import _capi as capi
from . import core

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

_ctx_cache = {}


def _ctx():
    """One C-ABI context per (device, torch stream)."""
    dev = torch.cuda.current_device()
    stream = torch.cuda.current_stream().cuda_stream
    key = (dev, stream)
    if key not in _ctx_cache:
        _ctx_cache[key] = capi.Context(dev, stream)
    return _ctx_cache[key]


def _is_cuda(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor) and x.is_cuda


class XYZLut:
    """core.XYZLut / core.XYZLutFloat with an HBM path for CUDA tensors."""

    def __init__(self, info, use_extrinsics: bool = True, dtype=np.float64):
        self.info = info
        self.use_extrinsics = use_extrinsics
        self.dtype = np.dtype(dtype)
        self._host = (core.XYZLut if self.dtype == np.float64 else core.XYZLutFloat)(info, use_extrinsics)
        self._dev = {}

    @property
    def direction(self):
        return self._host.direction

    @property
    def offset(self):
        return self._host.offset

    def _device_lut(self):
        ctx = _ctx()
        if id(ctx) not in self._dev:
            info = self.info
            tf = np.array(info.lidar_to_sensor_transform, dtype=np.float64)
            if self.use_extrinsics:  # xyzlut.cpp:91-106: extrinsics are in metres, the LUT in mm
                ext = np.array(info.sensor_to_body, dtype=np.float64)
                ext[:3, 3] /= 0.001
                tf = ext @ tf
            self._dev[id(ctx)] = (ctx, capi.Lut.from_calib(
                ctx, info.w, info.h, 0.001, np.array(info.beam_to_lidar_transform, dtype=np.float64), tf,
                info.beam_azimuth_angles, info.beam_altitude_angles))
        return self._dev[id(ctx)]

    def __call__(self, scan_or_range):
        if hasattr(scan_or_range, "field"):  # a LidarFrame / LidarScan
            return self._host(scan_or_range)
        if not _is_cuda(scan_or_range):
            return self._host(scan_or_range)
        rng = scan_or_range
        if rng.dim() != 2 or rng.shape[0] != self.info.h or rng.shape[1] != self.info.w:
            raise ValueError("unexpected image dimensions")
        if rng.dtype != torch.uint32:
            rng = rng.to(torch.int64).to(torch.uint32) if rng.dtype != torch.int32 else rng.view(torch.uint32)
        rng = rng.contiguous()
        ctx, lut = self._device_lut()
        tdt = torch.float64 if self.dtype == np.float64 else torch.float32
        out = torch.empty((self.info.h, self.info.w, 3), dtype=tdt, device=rng.device)
        capi.check(ctx.L.ouster_hip_cartesian(ctx.h, lut.h, rng.data_ptr(), out.data_ptr(),
                                              capi.F64 if tdt == torch.float64 else capi.F32, 1))
        return out


def XYZLutFloat(info, use_extrinsics: bool = True):
    return XYZLut(info, use_extrinsics, dtype=np.float32)


def destagger(info, field, inverse: bool = False):
    """core.destagger for numpy; the same kernel on HBM for CUDA tensors of any dtype/trailing dims."""
    if not _is_cuda(field):
        return core.destagger(info, field, inverse)
    if field.dim() < 2:
        raise ValueError("Expected at least two dimensions for destagger")
    shifts = np.ascontiguousarray(info.format.pixel_shift_by_row, dtype=np.int32)
    h, w = int(field.shape[0]), int(field.shape[1])
    if h != info.h or w != info.w or h != shifts.size or field.numel() == 0:
        raise ValueError("Image resolution must match SensorInfo.")
    src = field.contiguous()
    extra = int(np.prod(src.shape[2:])) if src.dim() > 2 else 1
    out = torch.empty_like(src)
    ctx = _ctx()
    capi.check(ctx.L.ouster_hip_destagger(ctx.h, src.data_ptr(), out.data_ptr(), h, w,
                                          src.element_size() * extra, shifts.ctypes.data, shifts.size,
                                          int(inverse), 1))
    return out
