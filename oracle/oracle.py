"""ctypes binding for the CPU ORACLE (test infrastructure, NOT product code).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  It wraps ``oracle/_build/libouster_oracle.so``
(built from ``ouster_oracle.c`` by ``oracle/Makefile``) and adds the small
host-side helpers the tests need: a classic-pcap reader, a metadata-JSON
calibration reader and a synthetic-frame generator.

Reference behaviour restated (paths relative to /root/reference):
  * pcap framing: SURVEY.md appendix B (classic pcap, Ethernet/IPv4/UDP)
  * metadata defaults: ouster_core/src/metadata.cpp:54-55,725-771,
    ouster_core/src/sensor_info.cpp:89-105
"""
from __future__ import annotations

import ctypes as C
import json
import os
import struct
import subprocess
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libouster_oracle.so")

MAX_FIELDS = 32
NAME_LEN = 24

VOID, U8, U16, U32, U64, I8, I16, I32, I64, F32, F64, CHAR, F16 = range(13)
NP_OF_TAG = {U8: np.uint8, U16: np.uint16, U32: np.uint32, U64: np.uint64,
             I8: np.int8, I16: np.int16, I32: np.int32, I64: np.int64,
             F32: np.float32, F64: np.float64, F16: np.uint16}

PROFILES = {
    "LEGACY": 1,
    "RNG19_RFL8_SIG16_NIR16_DUAL": 2,
    "RNG19_RFL8_SIG16_NIR16": 3,
    "RNG15_RFL8_NIR8": 4,
    "FIVE_WORD_PIXEL": 5,
    "FUSA_RNG15_RFL8_NIR8_DUAL": 6,
    "RNG15_RFL8_NIR8_DUAL": 7,
    "RNG15_RFL8_NIR8_ZONE16": 8,
    "RNG19_RFL8_SIG16_NIR16_ZONE16": 9,
    "RNG15_RFL8_WIN8": 10,
    "RNG19_RFL8_SIG16_ZONE16_DUAL": 11,
    "RNG19_RFL8_SIG16_NIR16_RGB16": 12,
    "RNG19_RFL8_SIG16_NIR16_RGB16_DUAL": 13,
}
HEADER_STANDARD, HEADER_FUSA = 0, 1


class FDI(C.Structure):
    _fields_ = [("ty_tag", C.c_int32), ("shift", C.c_int32),
                ("num_elements", C.c_int32), ("pad_", C.c_int32),
                ("offset", C.c_uint64), ("mask", C.c_uint64)]


class Field(C.Structure):
    _fields_ = [("name", C.c_char * NAME_LEN), ("info", FDI)]


class PF(C.Structure):
    _fields_ = [("profile", C.c_int32), ("header_type", C.c_int32),
                ("pixels_per_column", C.c_uint32), ("columns_per_packet", C.c_uint32),
                ("columns_per_frame", C.c_uint32), ("max_frame_id", C.c_uint32),
                ("packet_header_size", C.c_uint64), ("col_header_size", C.c_uint64),
                ("channel_data_size", C.c_uint64), ("col_footer_size", C.c_uint64),
                ("packet_footer_size", C.c_uint64), ("col_size", C.c_uint64),
                ("lidar_packet_size", C.c_uint64),
                ("n_fields", C.c_int32), ("pad_", C.c_int32),
                ("fields", Field * MAX_FIELDS),
                ("packet_type_info", FDI), ("frame_id_info", FDI), ("init_id_info", FDI),
                ("prod_sn_info", FDI), ("alert_flags_info", FDI),
                ("countdown_thermal_shutdown_info", FDI),
                ("countdown_shot_limiting_info", FDI), ("thermal_shutdown_info", FDI),
                ("shot_limiting_info", FDI), ("col_status_info", FDI),
                ("col_timestamp_info", FDI), ("col_measurement_id_info", FDI)]

    def field_names(self) -> List[str]:
        return [self.fields[i].name.decode() for i in range(self.n_fields)]

    def field(self, name: str) -> FDI:
        for i in range(self.n_fields):
            if self.fields[i].name.decode() == name:
                return self.fields[i].info
        raise KeyError(name)


class Plane(C.Structure):
    _fields_ = [("name", C.c_char * NAME_LEN), ("ty_tag", C.c_int32),
                ("n_extra", C.c_int32), ("data", C.c_void_p)]


class FrameS(C.Structure):
    _fields_ = [("h", C.c_uint32), ("w", C.c_uint32), ("cpp", C.c_uint32),
                ("n_packets", C.c_uint32), ("n_planes", C.c_int32), ("pad_", C.c_int32),
                ("planes", Plane * MAX_FIELDS),
                ("timestamp", C.POINTER(C.c_uint64)),
                ("measurement_id", C.POINTER(C.c_uint16)),
                ("status", C.POINTER(C.c_uint32)),
                ("packet_timestamp", C.POINTER(C.c_uint64)),
                ("alert_flags", C.POINTER(C.c_uint8)),
                ("frame_id", C.c_int64), ("frame_status", C.c_uint64),
                ("shutdown_countdown", C.c_uint16), ("shot_limiting_countdown", C.c_uint16),
                ("pad2_", C.c_uint32)]


def build(force: bool = False) -> str:
    """Compile the oracle shared library if it is missing or stale."""
    src = [os.path.join(_HERE, "ouster_oracle.c"), os.path.join(_HERE, "ouster_oracle.h")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    u8p = C.POINTER(C.c_uint8)
    L.ora_field_info.argtypes = [C.c_uint64] * 5 + [C.POINTER(FDI)]
    L.ora_fdi_get.restype = C.c_uint64
    L.ora_fdi_get.argtypes = [C.POINTER(FDI), C.c_void_p]
    L.ora_fdi_set.argtypes = [C.POINTER(FDI), C.c_void_p, C.c_uint64]
    L.ora_value_mask.restype = C.c_uint64
    L.ora_value_mask.argtypes = [C.POINTER(FDI)]
    L.ora_type_size.restype = C.c_size_t
    L.ora_add_custom_profile.argtypes = [C.POINTER(Field), C.c_int, C.c_uint64,
                                         C.POINTER(C.c_int32)]
    L.ora_pf_init.argtypes = [C.POINTER(PF), C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                              C.c_uint32]
    L.ora_block_parsable.argtypes = [C.POINTER(PF)]
    for name, rt in [("ora_frame_id", C.c_uint32), ("ora_init_id", C.c_uint32),
                     ("ora_prod_sn", C.c_uint64), ("ora_packet_type", C.c_uint16),
                     ("ora_alert_flags", C.c_uint8),
                     ("ora_col_measurement_id", C.c_uint16),
                     ("ora_col_timestamp", C.c_uint64), ("ora_col_status", C.c_uint32),
                     ("ora_col_encoder", C.c_uint32), ("ora_col_frame_id", C.c_uint16)]:
        fn = getattr(L, name)
        fn.restype = rt
        fn.argtypes = [C.POINTER(PF), C.c_void_p]
    L.ora_frame_id_difference.argtypes = [C.POINTER(PF), C.c_uint32, C.c_uint32]
    L.ora_col_field.argtypes = [C.POINTER(PF), C.c_void_p, C.c_char_p, C.c_void_p,
                                C.c_size_t, C.c_int]
    L.ora_block_field.argtypes = [C.POINTER(PF), C.c_void_p, C.c_size_t, C.c_int,
                                  C.c_char_p, C.c_void_p, C.c_int]
    L.ora_set_block.argtypes = [C.POINTER(PF), C.c_void_p, C.c_size_t, C.c_int,
                                C.c_char_p, C.c_void_p]
    L.ora_crc64.restype = C.c_uint64
    L.ora_crc64.argtypes = [C.c_void_p, C.c_size_t]
    L.ora_frame_new.restype = C.POINTER(FrameS)
    L.ora_frame_new.argtypes = [C.c_uint32] * 3
    L.ora_frame_free.argtypes = [C.POINTER(FrameS)]
    L.ora_frame_add_plane.argtypes = [C.POINTER(FrameS), C.c_char_p, C.c_int, C.c_int]
    L.ora_frame_add_default_planes.argtypes = [C.POINTER(FrameS), C.c_int, C.c_int]
    L.ora_frame_plane.restype = C.c_void_p
    L.ora_frame_plane.argtypes = [C.POINTER(FrameS), C.c_char_p]
    L.ora_frame_fill.argtypes = [C.POINTER(FrameS), C.c_int]
    L.ora_batcher_new.restype = C.c_void_p
    L.ora_batcher_new.argtypes = [C.POINTER(PF), C.c_int64, C.c_uint32]
    L.ora_batcher_free.argtypes = [C.c_void_p]
    L.ora_batcher_reset.argtypes = [C.c_void_p]
    L.ora_batcher_force_col_path.argtypes = [C.c_void_p, C.c_int]
    L.ora_batcher_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64,
                                    C.POINTER(FrameS)]
    L.ora_batcher_finalize.argtypes = [C.c_void_p, C.POINTER(FrameS)]
    L.ora_batcher_dropped.restype = C.c_uint64
    L.ora_batcher_dropped.argtypes = [C.c_void_p]
    L.ora_frame_to_packets.argtypes = [C.POINTER(FrameS), C.POINTER(PF), C.c_uint32,
                                       C.c_uint64, C.c_void_p, C.c_void_p]
    L.ora_destagger.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    L.ora_make_xyz_lut.argtypes = [C.c_size_t, C.c_size_t, C.c_double, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.c_void_p, C.c_void_p]
    for n in ("ora_cartesian_f64", "ora_cartesian_f32", "ora_cartesian_f64_omp",
              "ora_cartesian_f32_omp"):
        getattr(L, n).argtypes = [C.c_void_p] * 4 + [C.c_size_t]
        getattr(L, n).restype = None
    for n in ("ora_dewarp_f64", "ora_dewarp_f32"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
        getattr(L, n).restype = None
    for n in ("ora_dewarp_frame_f64", "ora_dewarp_frame_f32"):
        getattr(L, n).argtypes = [C.c_void_p] * 9 + [C.c_size_t, C.c_size_t, C.c_double, C.c_double]
        getattr(L, n).restype = C.c_size_t
    L.ora_bench_hot_path.restype = C.c_double
    L.ora_bench_hot_path.argtypes = [C.POINTER(PF), C.c_int, C.c_void_p, C.c_uint32, C.c_uint32,
                                     C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    L.ora_bench_hot_path2.restype = C.c_double
    L.ora_bench_hot_path2.argtypes = L.ora_bench_hot_path.argtypes + [C.c_int]
    L.ora_bench_stream_copy.restype = C.c_double
    L.ora_bench_stream_copy.argtypes = [C.c_size_t, C.c_int, C.c_int]
    _lib = L
    return L


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


# --------------------------------------------------------------------------- #
# packet format
# --------------------------------------------------------------------------- #
def packet_format(profile, h: int, cpp: int, w: int, header_type: int = HEADER_STANDARD) -> PF:
    if isinstance(profile, str):
        profile = PROFILES[profile]
    pf = PF()
    rc = lib().ora_pf_init(C.byref(pf), profile, header_type, h, cpp, w)
    if rc:
        raise ValueError(f"ora_pf_init failed ({rc})")
    return pf


def field_info(bit_start, bit_size, upshift=0, max_length=0, num_elements=1) -> FDI:
    f = FDI()
    rc = lib().ora_field_info(bit_start, bit_size, upshift, max_length, num_elements,
                              C.byref(f))
    if rc:
        raise ValueError(f"field_info failed ({rc})")
    return f


def add_custom_profile(fields: List[Tuple[str, Tuple[int, int, int, int]]],
                       chan_data_size: int) -> int:
    """fields: [(name, (ty_tag, offset, mask, shift))] as in
    tests/frame_batcher_test.cpp:676-704."""
    arr = (Field * len(fields))()
    for i, (name, (ty, off, mask, shift)) in enumerate(fields):
        arr[i].name = name.encode()
        arr[i].info.ty_tag = ty
        arr[i].info.offset = off
        arr[i].info.mask = mask
        arr[i].info.shift = shift
        arr[i].info.num_elements = 1
    return lib().ora_add_custom_profile(arr, len(fields), chan_data_size, None)


def packet_field(pf: PF, name: str, packet: np.ndarray) -> np.ndarray:
    """python/src/cpp/client/packet.cpp:173-210 -- [H, cpp] array via col_field."""
    info = pf.field(name)
    dt = NP_OF_TAG[info.ty_tag]
    out = np.zeros((pf.pixels_per_column, pf.columns_per_packet), dtype=dt)
    es = out.itemsize
    for icol in range(pf.columns_per_packet):
        col = _ptr(packet) + pf.packet_header_size + icol * pf.col_size
        rc = lib().ora_col_field(C.byref(pf), col, name.encode(), _ptr(out) + icol * es, es,
                                 pf.columns_per_packet)
        assert rc == 0
    return out


def packet_header(pf: PF, which: str, packet: np.ndarray) -> np.ndarray:
    """python/src/cpp/client/packet.cpp:212-256."""
    L = lib()
    fn, dt = {"TIMESTAMP": (L.ora_col_timestamp, np.uint64),
              "ENCODER_COUNT": (L.ora_col_encoder, np.uint32),
              "MEASUREMENT_ID": (L.ora_col_measurement_id, np.uint16),
              "STATUS": (L.ora_col_status, np.uint32),
              "FRAME_ID": (L.ora_col_frame_id, np.uint16)}[which]
    out = np.zeros(pf.columns_per_packet, dtype=dt)
    for icol in range(pf.columns_per_packet):
        col = _ptr(packet) + pf.packet_header_size + icol * pf.col_size
        out[icol] = fn(C.byref(pf), col)
    return out


# --------------------------------------------------------------------------- #
# frames
# --------------------------------------------------------------------------- #
class Frame:
    """Owning wrapper over ora_frame; planes exposed as numpy views."""

    def __init__(self, h: int, w: int, cpp: int):
        self._f = lib().ora_frame_new(h, w, cpp)
        if not self._f:
            raise ValueError("Cannot construct LidarFrame with zero width or height")
        self.h, self.w, self.cpp = h, w, cpp

    def __del__(self):
        try:
            if self._f:
                lib().ora_frame_free(self._f)
                self._f = None
        except Exception:
            pass

    @classmethod
    def for_profile(cls, profile, h, w, cpp, with_window=False) -> "Frame":
        if isinstance(profile, str):
            profile = PROFILES[profile]
        fr = cls(h, w, cpp)
        if lib().ora_frame_add_default_planes(fr._f, profile, int(with_window)):
            raise ValueError("Unknown lidar udp profile")
        return fr

    @property
    def s(self) -> FrameS:
        return self._f.contents

    def add_plane(self, name: str, ty_tag: int, n_extra: int = 1):
        if lib().ora_frame_add_plane(self._f, name.encode(), ty_tag, n_extra):
            raise ValueError(f"cannot add plane {name}")

    def plane_names(self) -> List[str]:
        return [self.s.planes[i].name.decode() for i in range(self.s.n_planes)]

    def plane(self, name: str) -> np.ndarray:
        s = self.s
        for i in range(s.n_planes):
            p = s.planes[i]
            if p.name.decode() == name:
                dt = np.dtype(NP_OF_TAG[p.ty_tag])
                shape = (self.h, self.w) if p.n_extra == 1 else (self.h, self.w, p.n_extra)
                n = int(np.prod(shape))
                buf = (C.c_uint8 * (n * dt.itemsize)).from_address(p.data)
                return np.frombuffer(buf, dtype=dt).reshape(shape)
        raise KeyError(name)

    def _hdr(self, ptr, ctype, n, dt):
        buf = (ctype * n).from_address(C.addressof(ptr.contents))
        return np.frombuffer(buf, dtype=dt)

    @property
    def timestamp(self): return self._hdr(self.s.timestamp, C.c_uint64, self.w, np.uint64)
    @property
    def measurement_id(self): return self._hdr(self.s.measurement_id, C.c_uint16, self.w, np.uint16)
    @property
    def status(self): return self._hdr(self.s.status, C.c_uint32, self.w, np.uint32)
    @property
    def packet_timestamp(self): return self._hdr(self.s.packet_timestamp, C.c_uint64, self.s.n_packets, np.uint64)
    @property
    def alert_flags(self): return self._hdr(self.s.alert_flags, C.c_uint8, self.s.n_packets, np.uint8)
    @property
    def frame_id(self): return self.s.frame_id
    @frame_id.setter
    def frame_id(self, v): self.s.frame_id = v

    def fill(self, byte_value: int):
        lib().ora_frame_fill(self._f, byte_value)


class Batcher:
    def __init__(self, pf: PF, init_id: int = 0, expected_packets: Optional[int] = None):
        self.pf = pf
        if expected_packets is None:
            expected_packets = pf.columns_per_frame // pf.columns_per_packet
        self._b = lib().ora_batcher_new(C.byref(pf), init_id, expected_packets)

    def __del__(self):
        try:
            if self._b:
                lib().ora_batcher_free(self._b)
                self._b = None
        except Exception:
            pass

    def force_col_path(self, on: bool = True):
        lib().ora_batcher_force_col_path(self._b, int(on))

    def reset(self):
        lib().ora_batcher_reset(self._b)

    def batch(self, packet: np.ndarray, host_ts: int, frame: Frame) -> bool:
        packet = np.ascontiguousarray(packet, dtype=np.uint8)
        rc = lib().ora_batcher_batch(self._b, _ptr(packet), packet.size, host_ts, frame._f)
        if rc < 0:
            raise RuntimeError(f"batch failed ({rc})")
        return bool(rc)

    def finalize(self, frame: Frame):
        lib().ora_batcher_finalize(self._b, frame._f)

    @property
    def dropped(self) -> int:
        return lib().ora_batcher_dropped(self._b)


def frame_to_packets(frame: Frame, pf: PF, init_id: int = 0, prod_sn: int = 0
                     ) -> Tuple[np.ndarray, np.ndarray]:
    n = frame.s.n_packets
    out = np.zeros((n, pf.lidar_packet_size), dtype=np.uint8)
    ts = np.zeros(n, dtype=np.uint64)
    k = lib().ora_frame_to_packets(frame._f, C.byref(pf), init_id, prod_sn, _ptr(out), _ptr(ts))
    if k < 0:
        raise ValueError("Mismatch between expected number of packets and "
                         "PacketFormat.columns_per_packet")
    return out[:k], ts[:k]


def destagger(img: np.ndarray, shifts, inverse: bool = False) -> np.ndarray:
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    es = img.itemsize * int(np.prod(img.shape[2:], dtype=np.int64))
    sh = np.ascontiguousarray(shifts, dtype=np.int32)
    out = np.empty_like(img)
    rc = lib().ora_destagger(_ptr(img), _ptr(out), h, w, es, _ptr(sh), sh.size, int(inverse))
    if rc:
        raise ValueError("image height does not match shifts size")
    return out


def make_xyz_lut(w, h, range_unit, beam_to_lidar, transform, az_deg, alt_deg
                 ) -> Tuple[np.ndarray, np.ndarray]:
    b2l = np.ascontiguousarray(beam_to_lidar, dtype=np.float64).reshape(4, 4)
    tf = np.ascontiguousarray(transform, dtype=np.float64).reshape(4, 4)
    az = np.ascontiguousarray(az_deg, dtype=np.float64)
    alt = np.ascontiguousarray(alt_deg, dtype=np.float64)
    if az.size != alt.size:
        raise ValueError("unexpected frame dimensions")
    d = np.zeros((w * h, 3), dtype=np.float64)
    o = np.zeros((w * h, 3), dtype=np.float64)
    rc = lib().ora_make_xyz_lut(w, h, range_unit, _ptr(b2l), _ptr(tf), _ptr(az), _ptr(alt),
                                az.size, _ptr(d), _ptr(o))
    if rc == -1:
        raise ValueError("lut dimensions must be greater than zero")
    if rc:
        raise ValueError("unexpected frame dimensions")
    return d, o


def cartesian(range_img: np.ndarray, direction: np.ndarray, offset: np.ndarray) -> np.ndarray:
    r = np.ascontiguousarray(range_img, dtype=np.uint32)
    n = r.size
    if n != direction.shape[0]:
        raise ValueError("unexpected image dimensions")
    if direction.dtype == np.float32:
        d = np.ascontiguousarray(direction); o = np.ascontiguousarray(offset, dtype=np.float32)
        pts = np.empty((n, 3), dtype=np.float32)
        lib().ora_cartesian_f32(_ptr(pts), _ptr(r), _ptr(d), _ptr(o), n)
    else:
        d = np.ascontiguousarray(direction, dtype=np.float64)
        o = np.ascontiguousarray(offset, dtype=np.float64)
        pts = np.empty((n, 3), dtype=np.float64)
        lib().ora_cartesian_f64(_ptr(pts), _ptr(r), _ptr(d), _ptr(o), n)
    return pts


def dewarp(points: np.ndarray, poses: np.ndarray, h: int, w: int) -> np.ndarray:
    """pose_util.h:38-56 -- points [h*w, 3], poses [w, 4, 4] (or [w, 16]) float64."""
    pts = np.ascontiguousarray(points)
    po = np.ascontiguousarray(poses, dtype=np.float64).reshape(w, 16)
    out = np.empty_like(pts)
    fn = lib().ora_dewarp_f32 if pts.dtype == np.float32 else lib().ora_dewarp_f64
    fn(_ptr(out), _ptr(pts), _ptr(po), h, w)
    return out


def dewarp_frame(range_img: np.ndarray, status: np.ndarray, timestamp: np.ndarray, poses: np.ndarray,
                 lut_dir: np.ndarray, lut_ofs: np.ndarray, min_range: float, max_range: float):
    """impl/dewarp_impl.h:23-81 -- range-gated, compacting dewarp of one frame.  The LUT dtype
    (float32 / float64) selects T.  Returns (points [n,3] T, col_idxs [n] u32, timestamps [n] u64)."""
    h, w = range_img.shape
    T = np.float32 if lut_dir.dtype == np.float32 else np.float64
    r = np.ascontiguousarray(range_img, dtype=np.uint32)
    st = np.ascontiguousarray(status, dtype=np.uint32)
    tsn = np.ascontiguousarray(timestamp, dtype=np.uint64)
    po = np.ascontiguousarray(poses, dtype=np.float64).reshape(w, 16)
    d = np.ascontiguousarray(lut_dir, dtype=T).reshape(h * w, 3)
    o = np.ascontiguousarray(lut_ofs, dtype=T).reshape(h * w, 3)
    out = np.empty((h * w, 3), dtype=T)
    col = np.empty(h * w, dtype=np.uint32)
    ts = np.empty(h * w, dtype=np.uint64)
    fn = lib().ora_dewarp_frame_f32 if T == np.float32 else lib().ora_dewarp_frame_f64
    n = fn(_ptr(out), _ptr(col), _ptr(ts), _ptr(r), _ptr(st), _ptr(tsn), _ptr(po), _ptr(d), _ptr(o),
           h, w, float(min_range), float(max_range))
    return out[:n].copy(), col[:n].copy(), ts[:n].copy()


# --------------------------------------------------------------------------- #
# calibration / metadata (minimal subset of SensorInfo)
# --------------------------------------------------------------------------- #
DEFAULT_LIDAR_TO_SENSOR = np.array([[-1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 1, 36.18],
                                    [0, 0, 0, 1]], dtype=np.float64)


@dataclass
class Calib:
    h: int
    w: int
    cpp: int
    profile: int
    header_type: int
    pixel_shift_by_row: np.ndarray
    beam_altitude_angles: np.ndarray
    beam_azimuth_angles: np.ndarray
    beam_to_lidar: np.ndarray
    lidar_to_sensor: np.ndarray
    extrinsic: np.ndarray = field(default_factory=lambda: np.eye(4))
    init_id: int = 0
    prod_sn: int = 0
    fw: Tuple[int, int, int] = (0, 0, 0)
    prod_line: str = ""

    def packet_format(self) -> PF:
        return packet_format(self.profile, self.h, self.cpp, self.w, self.header_type)

    @property
    def with_window(self) -> bool:
        # lidar_frame.cpp:1097-1110
        return self.fw >= (3, 2, 0)

    def lut_transform(self, use_extrinsics: bool) -> np.ndarray:
        # xyzlut.cpp:91-102
        tf = self.lidar_to_sensor.copy()
        if use_extrinsics:
            ext = self.extrinsic.copy()
            ext[:3, 3] /= 0.001
            tf = ext @ self.lidar_to_sensor
        return tf

    def xyz_lut(self, use_extrinsics: bool = False):
        return make_xyz_lut(self.w, self.h, 0.001, self.beam_to_lidar,
                            self.lut_transform(use_extrinsics),
                            self.beam_azimuth_angles, self.beam_altitude_angles)


def _default_b2l_x(prod_line: str) -> float:
    # sensor_info.cpp:89-98
    if prod_line.startswith("OS-0-"):
        return 27.67
    if prod_line.startswith("OS-1-"):
        return 15.806
    if prod_line.startswith("OS-2-"):
        return 13.762
    return 12.163


def _parse_fw(build_rev: str) -> Tuple[int, int, int]:
    import re
    m = re.search(r"v?(\d+)\.(\d+)\.(\d+)", build_rev or "")
    return tuple(int(x) for x in m.groups()) if m else (0, 0, 0)


def calib_from_json(path: str) -> Calib:
    d = json.load(open(path))
    if "lidar_data_format" in d:  # nested (fw >= 3) layout
        df = d["lidar_data_format"]
        bi = d["beam_intrinsics"]
        si = d.get("sensor_info", {})
        li = d.get("lidar_intrinsics", {})
        alt, az = bi["beam_altitude_angles"], bi["beam_azimuth_angles"]
        b2l = np.array(bi["beam_to_lidar_transform"], dtype=np.float64).reshape(4, 4) \
            if "beam_to_lidar_transform" in bi else None
        origin = bi.get("lidar_origin_to_beam_origin_mm")
        l2s = li.get("lidar_to_sensor_transform")
        prod_line = si.get("prod_line", "")
        init_id = si.get("initialization_id", 0)
        prod_sn = si.get("prod_sn", "0")
        build_rev = si.get("build_rev", "")
        header_type = d.get("config_params", {}).get("header_type")
    else:
        df = d.get("data_format", {})
        alt, az = d["beam_altitude_angles"], d["beam_azimuth_angles"]
        b2l = None
        origin = d.get("lidar_origin_to_beam_origin_mm")
        l2s = d.get("lidar_to_sensor_transform")
        prod_line = d.get("prod_line", "")
        init_id = d.get("initialization_id", 0)
        prod_sn = d.get("prod_sn", "0")
        build_rev = d.get("build_rev", "")
        header_type = None
    mode = d.get("lidar_mode") or d.get("config_params", {}).get("lidar_mode", "1024x10")
    w_mode = int(mode.split("x")[0])
    h = int(df.get("pixels_per_column", 64))  # data_format.cpp:79-125 defaults
    w = int(df.get("columns_per_frame", w_mode))
    cpp = int(df.get("columns_per_packet", 16))
    profile = PROFILES[df.get("udp_profile_lidar", "LEGACY")]
    shifts = df.get("pixel_shift_by_row")
    if shifts is None:
        shifts = [x * (w // 1024) if w >= 1024 else x // 2 for x in (18, 12, 6, 0)] * 16
    shifts = (list(shifts) + [0] * h)[:h]  # metadata.cpp:530-534
    if b2l is None:
        b2l = np.eye(4)
        b2l[0, 3] = origin if origin is not None else _default_b2l_x(prod_line)
    l2s = np.array(l2s, dtype=np.float64).reshape(4, 4) if l2s else DEFAULT_LIDAR_TO_SENSOR.copy()
    if header_type is None:  # metadata.cpp:545-555
        ht = HEADER_FUSA if profile == PROFILES["FUSA_RNG15_RFL8_NIR8_DUAL"] else HEADER_STANDARD
    else:
        ht = HEADER_FUSA if str(header_type).upper() == "FUSA" else HEADER_STANDARD
    ext = np.eye(4)
    sdk = d.get("ouster-sdk", {})
    if isinstance(sdk, dict) and sdk.get("extrinsic"):
        ext = np.array(sdk["extrinsic"], dtype=np.float64).reshape(4, 4)
    return Calib(h=h, w=w, cpp=cpp, profile=profile, header_type=ht,
                 pixel_shift_by_row=np.array(shifts, dtype=np.int32),
                 beam_altitude_angles=np.array(alt, dtype=np.float64),
                 beam_azimuth_angles=np.array(az, dtype=np.float64),
                 beam_to_lidar=b2l, lidar_to_sensor=l2s, extrinsic=ext,
                 init_id=int(init_id), prod_sn=int(prod_sn or 0), fw=_parse_fw(build_rev),
                 prod_line=prod_line)


def synthetic_calib(h=128, w=2048, cpp=16, profile="RNG15_RFL8_NIR8_DUAL",
                    b2l_x=13.762, extrinsic=None, header_type=HEADER_STANDARD) -> Calib:
    """OS-2-128 style calibration of SURVEY.md section 8(d)."""
    if isinstance(profile, str):
        profile = PROFILES[profile]
    alt = np.linspace(21.0, -21.0, h)
    az = np.tile(np.array([4.2, 1.4, -1.4, -4.2]), (h + 3) // 4)[:h]
    shifts = np.round(az / 360.0 * w).astype(np.int32)
    b2l = np.eye(4)
    b2l[0, 3] = b2l_x
    return Calib(h=h, w=w, cpp=cpp, profile=profile, header_type=header_type,
                 pixel_shift_by_row=shifts, beam_altitude_angles=alt,
                 beam_azimuth_angles=az, beam_to_lidar=b2l,
                 lidar_to_sensor=DEFAULT_LIDAR_TO_SENSOR.copy(),
                 extrinsic=np.eye(4) if extrinsic is None else np.asarray(extrinsic, float),
                 init_id=0x123456, prod_sn=0x1122334455, fw=(3, 2, 0))


# --------------------------------------------------------------------------- #
# classic pcap reader (Ethernet / IPv4 / UDP, no fragmentation)
# --------------------------------------------------------------------------- #
def read_pcap_udp(path: str) -> Iterator[Tuple[int, int, bytes]]:
    """Yield (timestamp_ns, dst_port, payload) per UDP datagram."""
    data = open(path, "rb").read()
    if len(data) < 24:
        return
    magic = struct.unpack_from("<I", data, 0)[0]
    if magic == 0xA1B2C3D4:
        endian, ns = "<", False
    elif magic == 0xA1B23C4D:
        endian, ns = "<", True
    elif magic == 0xD4C3B2A1:
        endian, ns = ">", False
    else:
        raise ValueError("not a classic pcap")
    linktype = struct.unpack_from(endian + "I", data, 20)[0]
    off = 24
    while off + 16 <= len(data):
        sec, frac, incl, _orig = struct.unpack_from(endian + "IIII", data, off)
        off += 16
        rec = data[off:off + incl]
        off += incl
        if linktype == 1:
            if len(rec) < 14:
                continue
            ethertype = struct.unpack_from(">H", rec, 12)[0]
            ip = rec[14:]
            if ethertype == 0x8100:
                ethertype = struct.unpack_from(">H", rec, 16)[0]
                ip = rec[18:]
            if ethertype != 0x0800:
                continue
        elif linktype == 113:  # linux cooked
            ip = rec[16:]
        else:
            continue
        if len(ip) < 20 or (ip[0] >> 4) != 4 or ip[9] != 17:
            continue
        ihl = (ip[0] & 0xF) * 4
        udp = ip[ihl:]
        if len(udp) < 8:
            continue
        dport, ulen = struct.unpack_from(">HH", udp, 2)
        payload = udp[8:ulen] if ulen >= 8 else udp[8:]
        ts = sec * 1_000_000_000 + (frac if ns else frac * 1000)
        yield ts, dport, bytes(payload)


def lidar_packets_from_pcap(path: str, pf: PF, port: int = 7502) -> np.ndarray:
    pk = [np.frombuffer(p, dtype=np.uint8) for _, dport, p in read_pcap_udp(path)
          if dport == port and len(p) == pf.lidar_packet_size]
    if not pk:
        return np.zeros((0, pf.lidar_packet_size), dtype=np.uint8)
    return np.stack(pk)


# --------------------------------------------------------------------------- #
# synthetic frames (cf. tests/packet_format_test.cpp:218-326)
# --------------------------------------------------------------------------- #
def randomize_frame(frame: Frame, pf: PF, seed: int, zero_range_frac: float = 0.3,
                    frame_id: int = 700, valid: bool = True):
    """Fill every plane named by the packet format with uniform values under its
    value mask (tests/util.h:84-96), headers like packet_format_test.cpp:238-259."""
    rng = np.random.default_rng(seed)
    names = set(pf.field_names())
    for name in frame.plane_names():
        if name not in names:
            continue
        pl = frame.plane(name)
        info = pf.field(name)
        mask = lib().ora_value_mask(C.byref(info))
        if pl.ndim == 3:  # RGB: three packed 16-bit words
            pl[...] = rng.integers(0, 0x10000, size=pl.shape, dtype=np.uint64).astype(pl.dtype)
            continue
        vals = rng.integers(0, mask + 1 if mask < 2 ** 63 else 2 ** 63, size=pl.shape,
                            dtype=np.uint64) & np.uint64(mask)
        if name in ("RANGE", "RANGE2") and zero_range_frac > 0:
            vals[rng.random(pl.shape) < zero_range_frac] = 0
        pl[...] = vals.astype(pl.dtype)
    w = frame.w
    frame.timestamp[:] = 1000 + np.arange(w, dtype=np.uint64)
    frame.measurement_id[:] = np.arange(w, dtype=np.uint16)
    legacy = pf.profile == PROFILES["LEGACY"]
    frame.status[:] = (0xFFFFFFFF if legacy else 1) if valid else 0
    frame.packet_timestamp[:] = 10 + np.arange(frame.s.n_packets, dtype=np.uint64)
    frame.alert_flags[:] = 0
    frame.frame_id = frame_id
    frame.s.frame_status = 0
    frame.s.shutdown_countdown = 0
    frame.s.shot_limiting_countdown = 0


def synth_packets(calib: Calib, n_frames: int, seed: int = 0xDEADBEEF,
                  zero_range_frac: float = 0.3, with_window: Optional[bool] = None
                  ) -> Tuple[np.ndarray, List[Frame]]:
    """n_frames synthetic frames -> ([n_frames, ppf, packet_size] uint8, source frames)."""
    pf = calib.packet_format()
    ww = calib.with_window if with_window is None else with_window
    ppf = calib.w // calib.cpp
    out = np.zeros((n_frames, ppf, pf.lidar_packet_size), dtype=np.uint8)
    frames = []
    for f in range(n_frames):
        fr = Frame.for_profile(calib.profile, calib.h, calib.w, calib.cpp, with_window=ww)
        randomize_frame(fr, pf, seed + f, zero_range_frac, frame_id=(700 + f) & 0xFFFF)
        pk, _ = frame_to_packets(fr, pf, calib.init_id & 0xFFFFFF, calib.prod_sn)
        assert pk.shape[0] == ppf
        out[f] = pk
        frames.append(fr)
    return out, frames


def batch_frame(pf: PF, packets: np.ndarray, frame: Frame, init_id: int = 0,
                host_ts: Optional[np.ndarray] = None, force_col: bool = False) -> bool:
    b = Batcher(pf, init_id=init_id)
    if force_col:
        b.force_col_path(True)
    done = False
    for i, p in enumerate(packets):
        ts = int(host_ts[i]) if host_ts is not None else 1 + i
        done = b.batch(p, ts, frame)
    return done
