"""ctypes face of oracle/_ref/libdewarp_ref.so -- the REFERENCE's own impl/dewarp_impl.h (range-gated, compacting frame
dewarp with provenance), compiled from where it lies by oracle/Makefile where /root/reference exists (the few ouster /
Eigen types it touches come from oracle/shims/ref_dewarp; Eigen3 is not in this image).  Test infrastructure: pins the
oracle's restatement ora_dewarp_frame_* and generates tests/golden/dewarp_ref_vectors.npz."""
import ctypes as C
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libdewarp_ref.so")
_lib = None


def available() -> bool:
    return os.path.exists(_PATH)


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_PATH)
        vp = C.c_void_p
        for n in ("ref_dewarp_frame_f64", "ref_dewarp_frame_f32"):
            f = getattr(_lib, n)
            f.restype = C.c_size_t
            f.argtypes = [vp] * 9 + [C.c_size_t, C.c_size_t, C.c_double, C.c_double]
        for n in ("ref_dewarp_frames_f64", "ref_dewarp_frames_f32"):
            f = getattr(_lib, n)
            f.restype = C.c_size_t
            f.argtypes = [vp] * 11 + [C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.c_double]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def dewarp_frame(range_img, status, timestamp, poses, lut_dir, lut_ofs, min_range, max_range):
    """dewarp_impl<T>(const LidarFrame&, ...) of the reference: (points [n,3] T, col_idxs u32, timestamps u64)."""
    h, w = range_img.shape
    T = np.float32 if lut_dir.dtype == np.float32 else np.float64
    r = np.ascontiguousarray(range_img, np.uint32)
    st = np.ascontiguousarray(status, np.uint32)
    ts = np.ascontiguousarray(timestamp, np.uint64)
    po = np.ascontiguousarray(poses, np.float64).reshape(w, 16)
    d = np.ascontiguousarray(lut_dir, T).reshape(h * w, 3)
    o = np.ascontiguousarray(lut_ofs, T).reshape(h * w, 3)
    out = np.empty((h * w, 3), T)
    col = np.empty(h * w, np.uint32)
    tn = np.empty(h * w, np.uint64)
    fn = _load().ref_dewarp_frame_f32 if T == np.float32 else _load().ref_dewarp_frame_f64
    n = fn(_p(out), _p(col), _p(tn), _p(r), _p(st), _p(ts), _p(po), _p(d), _p(o), h, w, float(min_range), float(max_range))
    return out[:n].copy(), col[:n].copy(), tn[:n].copy()


def dewarp_frames(ranges, statuses, timestamps, poses, lut_dir, lut_ofs, min_range, max_range, present=None):
    """dewarp_impl<T>(const FrameSet&, ...) of the reference over [n, h, w] ranges; present[f] == 0 leaves frame f out of
    the set (a null entry).  Returns (points, frame_idxs, col_idxs, timestamps)."""
    n_frames, h, w = ranges.shape
    T = np.float32 if lut_dir.dtype == np.float32 else np.float64
    r = np.ascontiguousarray(ranges, np.uint32)
    st = np.ascontiguousarray(statuses, np.uint32)
    ts = np.ascontiguousarray(timestamps, np.uint64)
    po = np.ascontiguousarray(poses, np.float64).reshape(n_frames, w, 16)
    d = np.ascontiguousarray(lut_dir, T).reshape(h * w, 3)
    o = np.ascontiguousarray(lut_ofs, T).reshape(h * w, 3)
    pr = np.ascontiguousarray(np.ones(n_frames) if present is None else present, np.uint8)
    cap = n_frames * h * w
    out = np.empty((cap, 3), T)
    fi = np.empty(cap, np.uint32)
    col = np.empty(cap, np.uint32)
    tn = np.empty(cap, np.uint64)
    fn = _load().ref_dewarp_frames_f32 if T == np.float32 else _load().ref_dewarp_frames_f64
    n = fn(_p(out), _p(fi), _p(col), _p(tn), _p(r), _p(st), _p(ts), _p(po), _p(d), _p(o), _p(pr), n_frames, h, w,
           float(min_range), float(max_range))
    return out[:n].copy(), fi[:n].copy(), col[:n].copy(), tn[:n].copy()
