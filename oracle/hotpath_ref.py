"""ctypes face of oracle/_ref/libhotpath_ref.so -- the timing harness (oracle/hotpath_ref.cpp, no reference code in it) that
drives the REFERENCE's own loops (libdecode_ref.so: block_field; libcore_ref.so: destagger_into, cartesianT) over a pool of
frames, on one thread or with the frames spread over OpenMP threads, and libcore_ref_omp.so (the same core library built with
-fopenmp -DOUSTER_OMP, the reference's own parallel cartesianT).  Test infrastructure: bench.py's cpu_baseline (kind
"reference") and tests/test_oracle_ref_hotpath.py; never used by the product."""
import ctypes as C
import os

import numpy as np

from . import core_ref, decode_ref

REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
PATH = os.path.join(REF, "libhotpath_ref.so")
CORE_OMP = os.path.join(REF, "libcore_ref_omp.so")
_lib = None


class Args(C.Structure):
    _fields_ = [("pf", C.c_void_p), ("packets", C.c_void_p), ("pool_frames", C.c_size_t), ("ppf", C.c_size_t),
                ("packet_stride", C.c_size_t), ("names", C.c_void_p), ("elem", C.c_void_p), ("n_planes", C.c_size_t),
                ("dst_idx", C.c_void_p), ("n_dst", C.c_size_t), ("xyz_idx", C.c_void_p), ("n_xyz", C.c_size_t),
                ("dir", C.c_void_p), ("ofs", C.c_void_p), ("h", C.c_size_t), ("w", C.c_size_t), ("shifts", C.c_void_p),
                ("block_dim", C.c_int), ("out_planes", C.c_void_p), ("out_cloud", C.c_void_p)]


def available() -> bool:
    return os.path.exists(PATH) and core_ref.available() and decode_ref.available()


def omp_available() -> bool:
    return available() and os.path.exists(CORE_OMP)


def lib():
    global _lib
    if _lib is None:
        decode_ref.lib()   # the handle of ref_pf_new lives in this instance; the harness dlopens the same file
        _lib = C.CDLL(PATH)
        _lib.ref_bench_hot_path.restype = C.c_double
        _lib.ref_bench_hot_path.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(Args), C.c_size_t, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(C.c_double)]
        _lib.ref_bench_cartesian_omp.restype = C.c_double
        _lib.ref_bench_cartesian_omp.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
    return _lib


class HotPath:
    """The reference's three loops over `pool` [frames, packets per frame, packet bytes] (uint8)."""

    def __init__(self, O, pf, pool: np.ndarray, plane_dtypes: dict, destaggered, xyz_fields, direction, offset, shifts, block_dim=16):
        self.rpf = decode_ref.RefPacketFormat(O, pf)
        self.pool = np.ascontiguousarray(pool)
        self.names = [n for n in self.rpf.names if n in plane_dtypes]
        self.dtypes = [np.dtype(plane_dtypes[n]) for n in self.names]
        self.h, self.w = int(pf.pixels_per_column), int(pf.columns_per_frame)
        self._names = (C.c_char_p * len(self.names))(*[n.encode() for n in self.names])
        self._elem = (C.c_size_t * len(self.names))(*[d.itemsize for d in self.dtypes])
        self._dst = (C.c_int * len(destaggered))(*[self.names.index(n) for n in destaggered])
        self._xyz = (C.c_int * len(xyz_fields))(*[self.names.index(n) for n in xyz_fields])
        self.dir = np.ascontiguousarray(direction, dtype=np.float64)
        self.ofs = np.ascontiguousarray(offset, dtype=np.float64)
        self.shifts = np.ascontiguousarray(shifts, dtype=np.int32)
        self.block_dim = block_dim
        self.n_dst, self.n_xyz = len(destaggered), len(xyz_fields)

    def run(self, n_frames: int, reps: int, threads: int = 1, own_inputs: bool = False, want_outputs: bool = False):
        """(wall seconds, (decode, destagger, cartesian) seconds of thread 0, planes of thread 0's last frame, its last cloud)."""
        a = Args()
        a.pf = self.rpf.h
        a.packets = self.pool.ctypes.data
        a.pool_frames, a.ppf, a.packet_stride = self.pool.shape[0], self.pool.shape[1], self.pool.strides[1]
        a.names = C.cast(self._names, C.c_void_p)
        a.elem = C.cast(self._elem, C.c_void_p)
        a.n_planes = len(self.names)
        a.dst_idx, a.n_dst = C.cast(self._dst, C.c_void_p), self.n_dst
        a.xyz_idx, a.n_xyz = C.cast(self._xyz, C.c_void_p), self.n_xyz
        a.dir, a.ofs = self.dir.ctypes.data, self.ofs.ctypes.data
        a.h, a.w = self.h, self.w
        a.shifts = self.shifts.ctypes.data
        a.block_dim = self.block_dim
        planes, cloud, keep = None, None, None
        if want_outputs:
            planes = {n: np.zeros((self.h, self.w), dtype=d) for n, d in zip(self.names, self.dtypes)}
            keep = (C.c_void_p * len(self.names))(*[planes[n].ctypes.data for n in self.names])
            a.out_planes = C.cast(keep, C.c_void_p)
            cloud = np.zeros((self.h * self.w, 3), dtype=np.float64)
            a.out_cloud = cloud.ctypes.data
        legs = (C.c_double * 3)()
        t = lib().ref_bench_hot_path(decode_ref.PATH.encode(), core_ref.PATH.encode(), C.byref(a), int(n_frames), int(reps),
                                     int(threads), 1 if own_inputs else 0, legs)
        if t < 0:
            raise RuntimeError("oracle/_ref: a reference library or symbol is missing")
        return t, tuple(legs), planes, cloud


def bench_cartesian_omp(range_img, direction, offset, reps: int, threads: int) -> float:
    """cartesianT<double> built with -DOUSTER_OMP (the reference's own parallel form), `reps` clouds: seconds."""
    r = np.ascontiguousarray(range_img, dtype=np.uint32)
    d = np.ascontiguousarray(direction, dtype=np.float64)
    o = np.ascontiguousarray(offset, dtype=np.float64)
    t = lib().ref_bench_cartesian_omp(CORE_OMP.encode(), r.ctypes.data, d.ctypes.data, o.ctypes.data, r.shape[0], r.shape[1],
                                      int(reps), int(threads))
    if t < 0:
        raise RuntimeError("oracle/_ref/libcore_ref_omp.so is missing")
    return t
