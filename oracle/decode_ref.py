"""ctypes face of oracle/_ref/libdecode_ref.so -- the REFERENCE's own packet-decode loop: PacketFormat::block_field<T, BlockDim>
(ouster_core/src/parsing.cpp:628-657) over FieldDecodeInfo::get<T> (include/ouster/core/field_decode_info.h:41-54), compiled
from where they lie by oracle/Makefile where /root/reference exists (oracle/decode_ref.cpp holds the stand-in PacketFormat with
the members that function reads).  Test infrastructure: pins the oracle's ora_block_field; bench.py times it as the CPU
baseline's decode leg."""
import ctypes as C
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libdecode_ref.so")
_lib = None


class RefFDI(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("mask", C.c_uint64), ("shift", C.c_int32), ("type_bytes", C.c_int32)]


def available() -> bool:
    return os.path.exists(PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(PATH)
        vp, sz = C.c_void_p, C.c_size_t
        _lib.ref_pf_new.restype = vp
        _lib.ref_pf_new.argtypes = [vp, C.POINTER(RefFDI)]
        _lib.ref_pf_add_field.restype = None
        _lib.ref_pf_add_field.argtypes = [vp, C.c_char_p, C.POINTER(RefFDI)]
        _lib.ref_pf_free.restype = None
        _lib.ref_pf_free.argtypes = [vp]
        _lib.ref_block_field.restype = C.c_int
        _lib.ref_block_field.argtypes = [vp, vp, sz, C.c_int, C.c_char_p, vp, C.c_int]
        _lib.ref_bench_decode_frame.restype = C.c_double
        _lib.ref_bench_decode_frame.argtypes = [vp, vp, sz, sz, vp, vp, vp, sz, C.c_int, C.c_int, C.c_int]
    return _lib


def _fdi(O, info) -> RefFDI:
    return RefFDI(int(info.offset), int(info.mask), int(info.shift), int(O.lib().ora_type_size(info.ty_tag)))


class RefPacketFormat:
    """The reference's block_field on the oracle's geometry and field tables (pf: oracle.PF)."""

    def __init__(self, O, pf):
        self.pf = pf
        geo = (C.c_uint64 * 6)(pf.packet_header_size, pf.col_header_size, pf.col_size, pf.channel_data_size,
                               pf.columns_per_packet, pf.pixels_per_column)
        mid = _fdi(O, pf.col_measurement_id_info)
        self.h = lib().ref_pf_new(geo, C.byref(mid))
        self.names = pf.field_names()
        for n in self.names:
            f = _fdi(O, pf.field(n))
            lib().ref_pf_add_field(self.h, n.encode(), C.byref(f))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_pf_free(self.h)
            self.h = None

    def block_field(self, data: np.ndarray, name: str, packet: np.ndarray, block_dim: int) -> int:
        """data [H, cols] is written at the packet's measurement ids; returns 0, or -2 where the reference throws."""
        assert data.flags.c_contiguous and packet.flags.c_contiguous
        return lib().ref_block_field(self.h, data.ctypes.data, data.itemsize, data.shape[1], name.encode(),
                                     packet.ctypes.data, block_dim)

    def bench_decode_frame(self, packets: np.ndarray, planes, block_dim: int, reps: int) -> float:
        """block_field of every plane in `planes` {name: [H, W] array} for every packet of the frame, `reps` times: seconds."""
        names = list(planes)
        arr_n = (C.c_char_p * len(names))(*[n.encode() for n in names])
        arr_p = (C.c_void_p * len(names))(*[planes[n].ctypes.data for n in names])
        arr_e = (C.c_size_t * len(names))(*[planes[n].itemsize for n in names])
        pk = np.ascontiguousarray(packets)
        return lib().ref_bench_decode_frame(self.h, pk.ctypes.data, pk.shape[0], pk.strides[0], arr_n, arr_p, arr_e, len(names),
                                            planes[names[0]].shape[1], block_dim, int(reps))
