"""ctypes binding of the C ABI in include/ouster_hip.h (plumbing for tests and bench.py).

The product is the C/C++ library pair built by the top-level Makefile:
  ouster_sdk_amd/lib/libouster_hip.so       HIP kernels + C ABI
  ouster_sdk_amd/lib/libouster_core_amd.so  C++ host API mirroring ouster::sdk::core
This module only loads them and mirrors the POD structs.  There is no CPU fallback: if the
HIP library is missing, or no GPU is visible when a context is requested, it raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
HIP_SO = os.environ.get("OUSTER_HIP_SO") or os.path.join(LIB_DIR, "libouster_hip.so")  # env: A/B builds
CORE_SO = os.path.join(LIB_DIR, "libouster_core_amd.so")

MAX_FIELDS = 32
U8, U16, U32, U64, F32, F64, F16 = 1, 2, 3, 4, 9, 10, 12

OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_RUNTIME = -2
ERR_NO_DEVICE = -3
ERR_UNSUPPORTED = -4


class Bits(C.Structure):
    _fields_ = [("mask", C.c_uint64), ("offset", C.c_uint32), ("shift", C.c_int32)]


class FieldDesc(C.Structure):
    _fields_ = [("bits", Bits), ("dst_elem_size", C.c_uint32), ("f16_nan_fill", C.c_uint32)]


class FormatDesc(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "pixels_per_column", "columns_per_packet", "columns_per_frame", "packet_header_size",
        "col_header_size", "channel_data_size", "col_footer_size", "packet_footer_size",
        "col_size", "lidar_packet_size")] + [(n, Bits) for n in (
            "col_timestamp", "col_measurement_id", "col_status", "frame_id", "alert_flags",
            "thermal_shutdown", "shot_limiting", "countdown_thermal_shutdown",
            "countdown_shot_limiting")] + [
        ("n_fields", C.c_uint32), ("reserved", C.c_uint32), ("fields", FieldDesc * MAX_FIELDS)]


class Calib(C.Structure):
    _fields_ = [("w", C.c_uint32), ("h", C.c_uint32), ("range_unit", C.c_double),
                ("beam_to_lidar_transform", C.c_double * 16), ("transform", C.c_double * 16),
                ("azimuth_angles_deg", C.POINTER(C.c_double)),
                ("altitude_angles_deg", C.POINTER(C.c_double)), ("n_angles", C.c_size_t)]


class FrameMeta(C.Structure):
    _fields_ = [("frame_id", C.c_int64), ("frame_status", C.c_uint64),
                ("shutdown_countdown", C.c_uint16), ("shot_limiting_countdown", C.c_uint16),
                ("n_valid_columns", C.c_uint32)]


class FrameOut(C.Structure):
    _fields_ = [("planes", C.c_void_p * MAX_FIELDS), ("destaggered", C.c_void_p * MAX_FIELDS),
                ("timestamp", C.c_void_p), ("measurement_id", C.c_void_p), ("status", C.c_void_p),
                ("packet_timestamp", C.c_void_p), ("alert_flags", C.c_void_p),
                ("frame_meta", C.c_void_p), ("xyz", C.c_void_p * 2), ("xyz_field", C.c_int32 * 2),
                ("xyz_dtype", C.c_int32), ("reserved", C.c_int32),
                ("gate_counts", C.c_void_p), ("gate_min_r", C.c_uint32), ("gate_max_r", C.c_uint32),
                ("gate_field", C.c_int32), ("reserved2", C.c_int32), ("xyz_poses", C.c_void_p)]


class OsfPlane(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("encoding", C.c_uint32), ("src_pixel_bytes", C.c_uint32),
                ("dst_elem_size", C.c_uint32), ("flags", C.c_uint32)]


# every symbol include/ouster_hip.h declares (checked by tests/test_abi.py)
class AllocStats(C.Structure):
    """ouster_hip_alloc_stats (include/ouster_hip.h)"""
    _fields_ = [(n, C.c_uint64) for n in ("device_allocs", "device_frees", "pinned_allocs", "pinned_frees", "pool_requests",
                                          "pool_hits", "pool_live_bytes", "pool_cached_bytes")]


ABI_SYMBOLS = [
    "ouster_hip_ctx_create", "ouster_hip_ctx_destroy", "ouster_hip_ctx_stream", "ouster_hip_ctx_device",
    "ouster_hip_ctx_set_knob", "ouster_hip_sync",
    "ouster_hip_last_error", "ouster_hip_version", "ouster_hip_format_create",
    "ouster_hip_format_destroy", "ouster_hip_lut_create", "ouster_hip_lut_create_from_arrays",
    "ouster_hip_lut_export", "ouster_hip_lut_destroy", "ouster_hip_decode", "ouster_hip_destagger",
    "ouster_hip_cartesian", "ouster_hip_osf_unpack", "ouster_hip_dewarp", "ouster_hip_dewarp_frames", "ouster_hip_dewarp_frames_counted", "ouster_hip_dewarp_frames_rows", "ouster_hip_range_gate", "ouster_hip_timing_enable", "ouster_hip_timing_read", "ouster_hip_last_decode_tile", "ouster_hip_last_decode_kernel",
    # host containers (round 6): pooled page-locked memory the GPU works on in place, frame-at-a-time calls on host arrays
    "ouster_hip_host_alloc", "ouster_hip_host_free", "ouster_hip_host_is_pinned", "ouster_hip_host_pool_trim",
    "ouster_hip_alloc_stats_read", "ouster_hip_device_alloc", "ouster_hip_device_free",
    "ouster_hip_destagger_host", "ouster_hip_cartesian_host", "ouster_hip_dewarp_host",
    "ouster_hip_copy_in", "ouster_hip_copy_out", "ouster_hip_ctx_scratch",
    "ouster_hip_ctx_set_tuning_cache", "ouster_hip_last_decode_tuner",
]

_hip = None
_core = None


class OusterHipError(RuntimeError):
    pass


def load_hip(private_path: Optional[str] = None):
    """Load libouster_hip.so (after torch, so both share one HIP runtime).
    private_path: an A/B build to load NEXT TO the product library in the same process (tools/ab/ab_inproc.py:
    same buffers, same physical placement for every variant); bound with RTLD_LOCAL | RTLD_DEEPBIND so that
    its internal calls stay inside it, not cached."""
    global _hip
    if private_path is None and _hip is not None:
        return _hip
    path = private_path or HIP_SO
    if not os.path.exists(path):
        raise OusterHipError(f"{path} is missing: run `make` (or __graft_entry__.build()) first; "
                             "there is no CPU fallback for the hot path")
    try:
        import torch  # noqa: F401  -- its bundled libamdhip64.so.7 must be the one in the process
    except Exception:
        pass
    if private_path is not None:
        L = C.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_DEEPBIND | os.RTLD_NOW)
    else:
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    L.ouster_hip_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.ouster_hip_ctx_destroy.argtypes = [vp]
    L.ouster_hip_ctx_stream.restype = vp
    L.ouster_hip_ctx_stream.argtypes = [vp]
    L.ouster_hip_sync.argtypes = [vp]
    if hasattr(L, "ouster_hip_ctx_set_knob"):   # absent only in older A/B builds loaded via OUSTER_HIP_SO
        L.ouster_hip_ctx_device.argtypes = [vp]
        L.ouster_hip_ctx_set_knob.argtypes = [vp, C.c_char_p, C.c_int]
    L.ouster_hip_last_error.restype = C.c_char_p
    L.ouster_hip_version.restype = C.c_char_p
    L.ouster_hip_format_create.argtypes = [vp, C.POINTER(FormatDesc), C.POINTER(vp)]
    L.ouster_hip_format_destroy.argtypes = [vp]
    L.ouster_hip_lut_create.argtypes = [vp, C.POINTER(Calib), C.POINTER(vp)]
    L.ouster_hip_lut_create_from_arrays.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_int,
                                                    C.POINTER(vp)]
    L.ouster_hip_lut_export.argtypes = [vp, vp, vp]
    L.ouster_hip_lut_destroy.argtypes = [vp]
    L.ouster_hip_decode.argtypes = [vp, vp, vp, C.c_size_t, C.c_uint32, vp, C.c_uint32, vp,
                                    C.POINTER(FrameOut), vp, vp, C.c_uint32]
    L.ouster_hip_destagger.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp,
                                       C.c_uint32, C.c_int, C.c_uint32]
    L.ouster_hip_cartesian.argtypes = [vp, vp, vp, vp, C.c_int, C.c_uint32]
    L.ouster_hip_dewarp.argtypes = [vp, vp, vp, vp, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]
    L.ouster_hip_dewarp_frames.argtypes = [vp, vp, C.c_uint32, vp, vp, vp, vp, C.c_uint32, C.c_double,
                                           C.c_double, C.c_int, vp, vp, vp, vp, C.c_uint64, vp]
    if hasattr(L, "ouster_hip_dewarp_frames_counted"):
        L.ouster_hip_dewarp_frames_counted.argtypes = [vp, vp, C.c_uint32, vp, vp, vp, vp, C.c_uint32, C.c_double,
                                                       C.c_double, C.c_int, vp, vp, vp, vp, C.c_uint64, vp, vp]
        L.ouster_hip_range_gate.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                            C.POINTER(C.c_int)]
    if hasattr(L, "ouster_hip_dewarp_frames_rows"):
        L.ouster_hip_dewarp_frames_rows.argtypes = [vp, vp, C.c_uint32, vp, vp, vp, vp, C.c_uint32, C.c_double, C.c_double,
                                                    vp, vp, vp, vp, C.c_uint64, vp, vp]
    L.ouster_hip_timing_enable.argtypes = [vp, C.c_int]
    L.ouster_hip_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.ouster_hip_last_decode_tile.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    if hasattr(L, "ouster_hip_last_decode_kernel"):   # absent only in older A/B builds loaded via OUSTER_HIP_SO
        L.ouster_hip_last_decode_kernel.restype = C.c_char_p
        L.ouster_hip_last_decode_kernel.argtypes = [vp]
    if hasattr(L, "ouster_hip_host_alloc"):   # absent only in older A/B builds loaded via OUSTER_HIP_SO
        L.ouster_hip_host_alloc.restype = vp
        L.ouster_hip_host_alloc.argtypes = [C.c_size_t, C.c_int]
        L.ouster_hip_host_free.argtypes = [vp]
        L.ouster_hip_host_free.restype = None
        L.ouster_hip_host_is_pinned.argtypes = [vp, C.c_size_t]
        L.ouster_hip_host_pool_trim.argtypes = [C.c_size_t]
        L.ouster_hip_host_pool_trim.restype = None
        L.ouster_hip_alloc_stats_read.argtypes = [C.POINTER(AllocStats)]
        L.ouster_hip_alloc_stats_read.restype = None
        L.ouster_hip_device_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        L.ouster_hip_device_free.argtypes = [vp]
        L.ouster_hip_device_free.restype = None
        L.ouster_hip_destagger_host.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint32, C.c_int]
        L.ouster_hip_cartesian_host.argtypes = [vp, vp, vp, vp, C.c_int]
        L.ouster_hip_dewarp_host.argtypes = [vp, vp, vp, vp, C.c_int, C.c_uint32, C.c_uint32]
        L.ouster_hip_copy_in.argtypes = [vp, vp, vp, C.c_size_t]
        L.ouster_hip_copy_out.argtypes = [vp, vp, vp, C.c_size_t]
        L.ouster_hip_ctx_scratch.argtypes = [vp, C.c_uint32, C.c_size_t, C.POINTER(vp)]
    if hasattr(L, "ouster_hip_ctx_set_tuning_cache"):
        L.ouster_hip_ctx_set_tuning_cache.argtypes = [vp, C.c_char_p]
        L.ouster_hip_last_decode_tuner.restype = C.c_char_p
        L.ouster_hip_last_decode_tuner.argtypes = [vp]
    if private_path is None:
        _hip = L
    return L


def alloc_stats() -> dict:
    """ouster_hip_alloc_stats_read as a dict (the library's hipMalloc / hipHostMalloc calls and the pool's hit count)."""
    st = AllocStats()
    load_hip().ouster_hip_alloc_stats_read(C.byref(st))
    return {n: int(getattr(st, n)) for n, _ in AllocStats._fields_}


def load_core():
    global _core
    if _core is not None:
        return _core
    load_hip()
    if not os.path.exists(CORE_SO):
        raise OusterHipError(f"{CORE_SO} is missing: run `make` first")
    L = C.CDLL(CORE_SO)
    L.ouster_core_format_desc.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.c_uint32,
                                          C.POINTER(FormatDesc), C.c_char_p, C.c_size_t]
    L.ouster_core_default_planes.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t,
                                             C.POINTER(C.c_uint32), C.c_uint32]
    _core = L
    return L


def check(rc: int):
    if rc == OK:
        return
    msg = load_hip().ouster_hip_last_error().decode(errors="replace")
    if rc == ERR_INVALID_ARGUMENT:
        raise ValueError(msg)
    raise OusterHipError(f"[{rc}] {msg}")


def default_planes(profile: str, with_window: bool = True) -> List[Tuple[str, int]]:
    buf = C.create_string_buffer(1024)
    sizes = (C.c_uint32 * MAX_FIELDS)()
    n = load_core().ouster_core_default_planes(profile.encode(), int(with_window), buf, 1024,
                                               sizes, MAX_FIELDS)
    if n < 0:
        raise ValueError("Unknown lidar udp profile")
    names = buf.value.decode().split(";") if n else []
    return [(names[i], int(sizes[i])) for i in range(n)]


def format_desc(profile: str, h: int, cpp: int, w: int, fields: Sequence[Tuple[str, int]],
                header_type: int = 0) -> FormatDesc:
    """FormatDesc for `profile` decoding the given (field name, element bytes) planes."""
    d = FormatDesc()
    names = (C.c_char_p * len(fields))(*[f[0].encode() for f in fields])
    sizes = (C.c_uint32 * len(fields))(*[int(f[1]) for f in fields])
    msg = C.create_string_buffer(256)
    rc = load_core().ouster_core_format_desc(profile.encode(), header_type, h, cpp, w, names,
                                             sizes, len(fields), C.byref(d), msg, 256)
    if rc:
        raise ValueError(msg.value.decode())
    return d


class Context:
    """Owns an ouster_hip_ctx.  stream: raw hipStream_t value (int; 0 = the null stream) or
    None to let the context create its own non-blocking stream."""

    STREAM_NULL = C.c_void_p(-1)  # OUSTER_HIP_STREAM_NULL

    def __init__(self, device: int = 0, stream: Optional[int] = None, lib=None):
        self.L = lib if lib is not None else load_hip()
        h = C.c_void_p()
        if stream is None:
            sp = None
        elif stream == 0:
            sp = self.STREAM_NULL
        else:
            sp = C.c_void_p(stream)
        check(self.L.ouster_hip_ctx_create(device, sp, C.byref(h)))
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.L.ouster_hip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.L.ouster_hip_sync(self.h))

    def set_knob(self, name: str, value: int):
        """Experiment / test knob of the context (see ouster_hip_ctx_set_knob)."""
        check(self.L.ouster_hip_ctx_set_knob(self.h, name.encode(), int(value)))

    def make_format(self, desc: FormatDesc) -> "Format":
        return Format(self, desc)

    def timing(self, on):
        """True / 1: HIP events around every decode kernel; N > 1: around every N-th (an event pair costs the stream 2 - 3 us)."""
        check(self.L.ouster_hip_timing_enable(self.h, int(on)))

    def last_decode_tile(self) -> Tuple[int, int]:
        c, r = C.c_int(), C.c_int()
        check(self.L.ouster_hip_last_decode_tile(self.h, C.byref(c), C.byref(r)))
        return c.value, r.value

    def last_decode_kernel(self) -> str:
        """Name of the optimistic-pass kernel the last decode launched (k_decode / k_decode_wide / k_decode_stream)."""
        if not hasattr(self.L, "ouster_hip_last_decode_kernel"):
            return ""
        return (self.L.ouster_hip_last_decode_kernel(self.h) or b"").decode()

    def set_tuning_cache(self, path: Optional[str]):
        """Persisted verdicts of the kernel-variant tuner (include/ouster_hip.h, ouster_hip_ctx_set_tuning_cache)."""
        check(self.L.ouster_hip_ctx_set_tuning_cache(self.h, path.encode() if path else None))

    def last_decode_tuner(self) -> str:
        """"cache" | "measured" | "measuring" | "none": how the last decode chose its kernel variant."""
        if not hasattr(self.L, "ouster_hip_last_decode_tuner"):
            return ""
        return (self.L.ouster_hip_last_decode_tuner(self.h) or b"").decode()

    def timing_read(self) -> Tuple[float, int]:
        ms = C.c_double()
        n = C.c_uint32()
        check(self.L.ouster_hip_timing_read(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


class Format:
    def __init__(self, ctx: Context, desc: FormatDesc):
        self.ctx = ctx
        self.desc = desc
        h = C.c_void_p()
        check(ctx.L.ouster_hip_format_create(ctx.h, C.byref(desc), C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if self.h:
                self.ctx.L.ouster_hip_format_destroy(self.h)
                self.h = None
        except Exception:
            pass


class Lut:
    def __init__(self, ctx: Context, handle):
        self.ctx = ctx
        self.h = handle

    @classmethod
    def from_calib(cls, ctx: Context, w, h, range_unit, beam_to_lidar, transform, az, alt) -> "Lut":
        import numpy as np
        az = np.ascontiguousarray(az, dtype=np.float64)
        alt = np.ascontiguousarray(alt, dtype=np.float64)
        c = Calib()
        c.w, c.h, c.range_unit = w, h, range_unit
        b = np.ascontiguousarray(beam_to_lidar, dtype=np.float64).reshape(16)
        t = np.ascontiguousarray(transform, dtype=np.float64).reshape(16)
        for i in range(16):
            c.beam_to_lidar_transform[i] = b[i]
            c.transform[i] = t[i]
        c.azimuth_angles_deg = az.ctypes.data_as(C.POINTER(C.c_double))
        c.altitude_angles_deg = alt.ctypes.data_as(C.POINTER(C.c_double))
        c.n_angles = min(az.size, alt.size) if az.size == alt.size else 0
        hd = C.c_void_p()
        check(ctx.L.ouster_hip_lut_create(ctx.h, C.byref(c), C.byref(hd)))
        return cls(ctx, hd)

    @classmethod
    def from_arrays(cls, ctx: Context, direction, offset, h, w) -> "Lut":
        import numpy as np
        d = np.ascontiguousarray(direction)
        o = np.ascontiguousarray(offset, dtype=d.dtype)
        dt = F32 if d.dtype == np.float32 else F64
        hd = C.c_void_p()
        check(ctx.L.ouster_hip_lut_create_from_arrays(ctx.h, d.ctypes.data, o.ctypes.data, h, w,
                                                      dt, C.byref(hd)))
        return cls(ctx, hd)

    def export(self, w, h):
        import numpy as np
        d = np.empty((w * h, 3), dtype=np.float64)
        o = np.empty((w * h, 3), dtype=np.float64)
        check(self.ctx.L.ouster_hip_lut_export(self.h, d.ctypes.data, o.ctypes.data))
        return d, o

    def __del__(self):
        try:
            if self.h:
                self.ctx.L.ouster_hip_lut_destroy(self.h)
                self.h = None
        except Exception:
            pass
