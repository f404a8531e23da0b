"""ctypes face of oracle/_ref/libzpng_ref.so -- the REFERENCE's own thirdparty/zpng/zpng.cpp, compiled by
oracle/Makefile where /root/reference exists.  Test infrastructure: pins the ZPNG restatement of
oracle/osf_oracle.py and generates the committed vectors of tests/golden/osf/zpng_ref_vectors.json."""
import ctypes
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libzpng_ref.so")


class _Buffer(ctypes.Structure):
    _fields_ = [("Data", ctypes.POINTER(ctypes.c_ubyte)), ("Bytes", ctypes.c_uint)]


class _ImageData(ctypes.Structure):
    _fields_ = [("Buffer", _Buffer), ("BytesPerChannel", ctypes.c_uint), ("Channels", ctypes.c_uint),
                ("WidthPixels", ctypes.c_uint), ("HeightPixels", ctypes.c_uint), ("StrideBytes", ctypes.c_uint)]


def available() -> bool:
    return os.path.exists(_PATH)


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_PATH)
        _lib.ZPNG_Compress.restype = _Buffer
        _lib.ZPNG_Compress.argtypes = [ctypes.POINTER(_ImageData)]
        _lib.ZPNG_Decompress.restype = _ImageData
        _lib.ZPNG_Decompress.argtypes = [_Buffer]
        _lib.ZPNG_Free.restype = None
        _lib.ZPNG_Free.argtypes = [ctypes.POINTER(_Buffer)]
    return _lib


# how zpng_lidarframe_encoder.cpp:52-73 presents a field plane to the codec
LAYOUT = {1: (1, 1), 2: (1, 2), 4: (4, 1), 8: (4, 2)}   # itemsize -> (channels, bytes per channel)


def compress(plane: np.ndarray) -> bytes:
    """ZPNG_Compress of an H x W plane of u8 / u16 / u32 / u64, laid out like the reference's OSF writer."""
    lib = _load()
    plane = np.ascontiguousarray(plane)
    h, w = plane.shape
    ch, bpc = LAYOUT[plane.dtype.itemsize]
    raw = plane.view(np.uint8).reshape(-1)
    img = _ImageData()
    img.Buffer.Data = raw.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))
    img.Buffer.Bytes = raw.size
    img.BytesPerChannel, img.Channels = bpc, ch
    img.WidthPixels, img.HeightPixels = w, h
    img.StrideBytes = w * ch * bpc
    out = lib.ZPNG_Compress(ctypes.byref(img))
    if not out.Data:
        raise RuntimeError("ZPNG_Compress failed")
    data = bytes(ctypes.cast(out.Data, ctypes.POINTER(ctypes.c_ubyte * out.Bytes)).contents)
    lib.ZPNG_Free(ctypes.byref(out))
    return data


def decompress(data: bytes):
    """ZPNG_Decompress -> (pixel bytes, width, height, channels, bytes per channel)."""
    lib = _load()
    src = (ctypes.c_ubyte * len(data)).from_buffer_copy(data)
    buf = _Buffer(ctypes.cast(src, ctypes.POINTER(ctypes.c_ubyte)), len(data))
    img = lib.ZPNG_Decompress(buf)
    if not img.Buffer.Data:
        raise RuntimeError("ZPNG_Decompress failed")
    px = bytes(ctypes.cast(img.Buffer.Data, ctypes.POINTER(ctypes.c_ubyte * img.Buffer.Bytes)).contents)
    res = (px, img.WidthPixels, img.HeightPixels, img.Channels, img.BytesPerChannel)
    lib.ZPNG_Free(ctypes.byref(img.Buffer))
    return res
