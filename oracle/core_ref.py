"""ctypes face of oracle/_ref/libcore_ref.so -- the REFERENCE's own cartesianT<T> (impl/cartesian.h:36-66) and
destagger_into<T> (impl/lidar_frame_impl.h:733-760), compiled from where they lie by oracle/Makefile where /root/reference
exists (oracle/shims/ref_core supplies the few Eigen / ouster names they touch).  Test infrastructure: pins the oracle's
restatements (ora_cartesian_*, ora_destagger); bench.py times it as the CPU baseline's reference legs."""
import ctypes as C
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libcore_ref.so")
_lib = None


def available() -> bool:
    return os.path.exists(PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(PATH)
        vp, sz = C.c_void_p, C.c_size_t
        for n in ("ref_cartesian_f64", "ref_cartesian_f32"):
            getattr(_lib, n).restype = None
            getattr(_lib, n).argtypes = [vp, vp, vp, vp, sz, sz]
        _lib.ref_destagger.restype = C.c_int
        _lib.ref_destagger.argtypes = [vp, vp, sz, sz, sz, vp, sz, C.c_int]
        _lib.ref_bench_frame_legs.restype = C.c_double
        _lib.ref_bench_frame_legs.argtypes = [vp, vp, sz, vp, sz, vp, vp, sz, sz, vp, C.c_int, vp, vp]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def cartesian(range_img, direction, offset):
    """cartesianT<T>(points, range, direction, offset), T = the LUT's dtype: [h*w, 3]."""
    r = np.ascontiguousarray(range_img, dtype=np.uint32)
    h, w = r.shape
    T = np.float32 if direction.dtype == np.float32 else np.float64
    d, o = np.ascontiguousarray(direction, dtype=T), np.ascontiguousarray(offset, dtype=T)
    pts = np.full((h * w, 3), np.nan, dtype=T)
    (lib().ref_cartesian_f32 if T == np.float32 else lib().ref_cartesian_f64)(_p(pts), _p(r), _p(d), _p(o), h, w)
    return pts


def destagger(img, shifts, inverse=False):
    """destagger_into<T>(img, pixel_shift_by_row, inverse, destaggered); ValueError where the reference throws."""
    img = np.ascontiguousarray(img)
    h, w = img.shape
    sh = np.ascontiguousarray(shifts, dtype=np.int32)
    out = np.empty_like(img)
    rc = lib().ref_destagger(_p(img), _p(out), h, w, img.itemsize, _p(sh), sh.size, int(inverse))
    if rc == -1:
        raise ValueError("image height does not match shifts size")
    assert rc == 0
    return out


def bench_frame_legs(dst_planes, range_planes, direction, offset, shifts, reps):
    """destagger_into<T> of every plane in dst_planes + cartesianT<double> of every range plane, `reps` times on one core:
    (seconds destagger, seconds cartesian)."""
    dst = [np.ascontiguousarray(p) for p in dst_planes]
    rng = [np.ascontiguousarray(p, dtype=np.uint32) for p in range_planes]
    h, w = rng[0].shape
    d, o = np.ascontiguousarray(direction, dtype=np.float64), np.ascontiguousarray(offset, dtype=np.float64)
    sh = np.ascontiguousarray(shifts, dtype=np.int32)
    pd = (C.c_void_p * len(dst))(*[p.ctypes.data for p in dst])
    es = (C.c_size_t * len(dst))(*[p.itemsize for p in dst])
    pr = (C.c_void_p * len(rng))(*[p.ctypes.data for p in rng])
    td, tc = C.c_double(), C.c_double()
    lib().ref_bench_frame_legs(pd, es, len(dst), pr, len(rng), _p(d), _p(o), h, w, _p(sh), int(reps), C.byref(td), C.byref(tc))
    return td.value, tc.value
