"""CPU oracle for SURVEY.md section 8 row f-4: OSF field-plane decode.  TEST INFRASTRUCTURE ONLY: nothing
under ouster_sdk_amd/ or include/ imports it; tests/ compare the product's OSF path against it.

Restates, in plain Python / numpy (zlib from the standard library, libzstd.so.1 through ctypes):
  file layout          ouster_osf/src/fb_utils.cpp:60-150, ouster_osf/fb/header.fbs, metadata.fbs, chunk.fbs
                       [u32 size][flatbuffer][u32 crc32 of both] blocks: header, chunks, metadata
  LidarScanMsg         ouster_osf/fb/os_sensor/lidar_scan_stream.fbs, restore_lidar_frame
                       ouster_osf/src/stream_lidar_frame.cpp:165-340
  decode_field         ouster_osf/src/png_tools.cpp:664-745: ZPNG first (never destaggered), else PNG by
                       bit depth (8 gray / 16 gray / 24 RGB / 32 RGBA / 64 RGBA16), then stagger()
  PNG sample order     ouster_osf/src/png_lidarframe_encoder.cpp:163-405 (png_set_swap: 16-bit samples are
                       the byte-swapped halves of the little-endian value), PNG spec filters 0-4
  ZPNG                 thirdparty/zpng/zpng.cpp:47-360, 490-600 (Zpng by C. A. Taylor, vendored by the
                       reference): 8-byte header {magic 0xFBF8, w, h, channels, bytes/channel}, zstd body,
                       per-row left-delta per byte; 3/4-byte pixels are split in colour planes with the
                       GB-RG transform
Pinned: tests/test_oracle_osf.py decodes the reference's own tests/osfs/OS-1-128_v2.3.0_1024x10_lb_n3.osf and
compares every plane and header with the frames the (golden-pinned) packet oracle batches from the pcap
the reference wrote that file from (tests/pcaps/OS-1-128_v2.3.0_1024x10_lb_n3.pcap); the ZPNG codec against the
reference's own zpng.cpp compiled into oracle/_ref/libzpng_ref.so (oracle/Makefile) and the vectors it produced
(tests/golden/osf/zpng_ref_vectors.json).
"""
from __future__ import annotations

import ctypes as C
import json
import struct
import zlib
from typing import Dict, List, Optional, Tuple

import numpy as np

# ---------------------------------------------------------------------------------------------
# minimal FlatBuffers reader (little endian, tables / vectors / strings / structs)
# ---------------------------------------------------------------------------------------------


class Table:
    def __init__(self, buf: bytes, pos: int):
        self.buf, self.pos = buf, pos
        self.vt = pos - struct.unpack_from("<i", buf, pos)[0]
        self.vt_len = struct.unpack_from("<H", buf, self.vt)[0]

    def _off(self, field: int) -> int:
        o = 4 + 2 * field
        if o + 2 > self.vt_len:
            return 0
        return struct.unpack_from("<H", self.buf, self.vt + o)[0]

    def scalar(self, field: int, fmt: str, default=0):
        o = self._off(field)
        return struct.unpack_from("<" + fmt, self.buf, self.pos + o)[0] if o else default

    def _indirect(self, field: int) -> Optional[int]:
        o = self._off(field)
        if not o:
            return None
        p = self.pos + o
        return p + struct.unpack_from("<I", self.buf, p)[0]

    def table(self, field: int) -> Optional["Table"]:
        p = self._indirect(field)
        return Table(self.buf, p) if p is not None else None

    def string(self, field: int) -> Optional[str]:
        p = self._indirect(field)
        if p is None:
            return None
        n = struct.unpack_from("<I", self.buf, p)[0]
        return self.buf[p + 4:p + 4 + n].decode()

    def vector(self, field: int, dtype) -> Optional[np.ndarray]:
        p = self._indirect(field)
        if p is None:
            return None
        n = struct.unpack_from("<I", self.buf, p)[0]
        return np.frombuffer(self.buf, dtype=dtype, count=n, offset=p + 4)

    def table_vector(self, field: int) -> List["Table"]:
        p = self._indirect(field)
        if p is None:
            return []
        n = struct.unpack_from("<I", self.buf, p)[0]
        out = []
        for i in range(n):
            e = p + 4 + 4 * i
            out.append(Table(self.buf, e + struct.unpack_from("<I", self.buf, e)[0]))
        return out


def size_prefixed_root(buf: bytes, pos: int) -> Tuple[Table, int]:
    """Root table of the size-prefixed flatbuffer at `pos`; also its body size."""
    size = struct.unpack_from("<I", buf, pos)[0]
    root = pos + 4 + struct.unpack_from("<I", buf, pos + 4)[0]
    return Table(buf, root), size


def block_crc_ok(buf: bytes, pos: int) -> bool:
    size = struct.unpack_from("<I", buf, pos)[0]
    stored = struct.unpack_from("<I", buf, pos + 4 + size)[0]
    return zlib.crc32(buf[pos:pos + 4 + size]) & 0xFFFFFFFF == stored


# ---------------------------------------------------------------------------------------------
# container
# ---------------------------------------------------------------------------------------------
CHAN_FIELD = {1: "RANGE", 2: "RANGE2", 3: "SIGNAL", 4: "SIGNAL2", 5: "REFLECTIVITY", 6: "REFLECTIVITY2",
              7: "NEAR_IR", 8: "FLAGS", 9: "FLAGS2", 40: "RAW_HEADERS", 45: "RAW32_WORD5", 46: "RAW32_WORD6",
              47: "RAW32_WORD7", 48: "RAW32_WORD8", 49: "RAW32_WORD9", 60: "RAW32_WORD1", 61: "RAW32_WORD2",
              62: "RAW32_WORD3", 63: "RAW32_WORD4", **{50 + i: f"CUSTOM{i}" for i in range(10)}}
FIELD_DTYPE = {1: np.uint8, 2: np.uint16, 3: np.uint32, 4: np.uint64}
# CHAN_FIELD_TYPE of os_sensor/common.fbs:4-19, as far as custom fields can carry them (encoded through uint views of the same size)
CUSTOM_DTYPE = {1: np.uint8, 2: np.uint16, 3: np.uint32, 4: np.uint64, 5: np.int8, 6: np.int16, 7: np.int32, 8: np.int64,
                9: np.float32, 10: np.float64, 11: "S1"}   # 11 = CHAR (field.cpp:62; e.g. POSITION_STRING): one byte per element


class OsfFile:
    def __init__(self, path: str):
        self.buf = open(path, "rb").read()
        hdr, hsize = size_prefixed_root(self.buf, 0)
        if self.buf[8:12] != b"OSF$":
            raise ValueError("not an OSF file")
        self.header_ok = block_crc_ok(self.buf, 0)
        self.version = hdr.scalar(0, "Q")
        self.status = hdr.scalar(1, "B")
        self.metadata_offset = hdr.scalar(2, "Q", 1)
        self.file_length = hdr.scalar(3, "Q", 1)
        self.chunks_base = 4 + hsize + 4
        meta, _ = size_prefixed_root(self.buf, self.metadata_offset)
        self.metadata_ok = block_crc_ok(self.buf, self.metadata_offset)
        self.id = meta.string(0)
        chunks = meta.vector(3, np.dtype([("start", "<u8"), ("end", "<u8"), ("offset", "<u8")]))
        self.chunk_offsets = [int(c["offset"]) for c in chunks] if chunks is not None else []
        self.entries: Dict[int, Tuple[str, bytes]] = {}
        for e in meta.table_vector(4):
            b = e.vector(2, np.uint8)
            self.entries[e.scalar(0, "I")] = (e.string(1), bytes(b) if b is not None else b"")

    def sensor_metadata(self) -> Dict[int, dict]:
        """id -> parsed sensor metadata JSON of every ouster/v1/os_sensor/LidarSensor entry."""
        out = {}
        for mid, (typ, b) in self.entries.items():
            if typ.endswith("LidarSensor"):
                t, _ = size_prefixed_root(b, 0)
                out[mid] = json.loads(t.string(0))
        return out

    def lidar_streams(self) -> Dict[int, int]:
        """stream id -> sensor metadata id."""
        out = {}
        for mid, (typ, b) in self.entries.items():
            if typ.endswith("LidarScanStream"):
                t, _ = size_prefixed_root(b, 0)
                out[mid] = t.scalar(0, "I")
        return out

    def messages(self):
        """(ts, stream id, message bytes) of every StampedMessage, chunk by chunk, sorted by ts."""
        out = []
        for off in self.chunk_offsets:
            pos = self.chunks_base + off
            if not block_crc_ok(self.buf, pos):
                raise ValueError("chunk crc mismatch")
            ch, _ = size_prefixed_root(self.buf, pos)
            for m in ch.table_vector(0):
                b = m.vector(2, np.uint8)
                out.append((m.scalar(0, "Q"), m.scalar(1, "I"), bytes(b)))
        return sorted(out, key=lambda x: x[0])


# ---------------------------------------------------------------------------------------------
# field codecs
# ---------------------------------------------------------------------------------------------
_zstd = None


def zstd_decompress(data: bytes, out_size: int) -> bytes:
    global _zstd
    if _zstd is None:
        _zstd = C.CDLL("libzstd.so.1")
        _zstd.ZSTD_decompress.restype = C.c_size_t
        _zstd.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        _zstd.ZSTD_isError.restype = C.c_uint
        _zstd.ZSTD_isError.argtypes = [C.c_size_t]
    out = C.create_string_buffer(out_size)
    n = _zstd.ZSTD_decompress(out, out_size, data, len(data))
    if _zstd.ZSTD_isError(n):
        raise ValueError("zstd error")
    return out.raw[:n]


def png_pixels(data: bytes) -> Tuple[np.ndarray, int, int, int, int]:
    """Unfiltered scanlines [h, w * bytes_per_pixel] of a non-interlaced PNG + (w, h, depth, colour)."""
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos, idat, ihdr = 8, [], None
    while pos + 8 <= len(data):
        n, typ = struct.unpack_from(">I4s", data, pos)
        body = data[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
        pos += 12 + n
    w, h, depth, colour, _, _, interlace = ihdr
    if interlace:
        raise ValueError("interlaced PNG")
    channels = {0: 1, 2: 3, 4: 2, 6: 4}[colour]
    bpp = channels * depth // 8
    stride = w * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 1:   # Sub: per byte lane a running sum mod 256
            cur = np.cumsum(line.reshape(-1, bpp), axis=0).reshape(-1) & 0xFF
        elif ft == 2:   # Up
            cur = (line + prev) & 0xFF
        else:           # Average / Paeth: sequential in x
            cur = np.zeros(stride, np.int32)
            for x in range(stride):
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    c = prev[x - bpp] if x >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + pred) & 0xFF
        out[y] = cur
        prev = cur
    return out, w, h, depth, colour


def decode_png_field(data: bytes, dtype, h: int, w: int) -> np.ndarray:
    px, pw, ph, depth, colour = png_pixels(data)
    if (pw, ph) != (w, h):
        raise ValueError("png size mismatch")
    bpp = px.shape[1] // w
    b = px.reshape(h, w, bpp).astype(np.uint64)
    if depth == 16:   # samples are the byte-swapped halves of the little-endian value (png_set_swap)
        b = b.reshape(h, w, bpp // 2, 2)[..., ::-1].reshape(h, w, bpp)
    val = np.zeros((h, w), np.uint64)
    for k in range(bpp):
        val |= b[..., k] << np.uint64(8 * k)
    return val.astype(dtype)


def decode_zpng_field(data: bytes, dtype, h: int, w: int) -> Optional[np.ndarray]:
    if len(data) < 8:
        return None
    magic, zw, zh, ch, bpc = struct.unpack_from("<HHHBB", data, 0)
    if magic != 0xFBF8:
        return None
    pixel_bytes = ch * bpc
    if (zw, zh) != (w, h) or pixel_bytes != np.dtype(dtype).itemsize:
        raise ValueError("Invalid allocation")
    body = np.frombuffer(zstd_decompress(data[8:], w * h * pixel_bytes), np.uint8)
    if pixel_bytes in (3, 4):     # colour planes + GB-RG transform, then the left delta
        planes = body.reshape(pixel_bytes, h, w).astype(np.int32)
        y, u, v = planes[0], planes[1], planes[2]
        B = y
        G = (u + B) & 0xFF
        r0 = (G - v) & 0xFF
        chans = [r0, G, B] + ([planes[3]] if pixel_bytes == 4 else [])
        out = np.stack([np.cumsum(c, axis=1) & 0xFF for c in chans], axis=-1).astype(np.uint8)
    else:
        out = (np.cumsum(body.reshape(h, w, pixel_bytes).astype(np.int32), axis=1) & 0xFF).astype(np.uint8)
    return np.ascontiguousarray(out).reshape(h, w * pixel_bytes).view(dtype).reshape(h, w)


def stagger(img: np.ndarray, shifts) -> np.ndarray:
    out = np.empty_like(img)
    for r in range(img.shape[0]):
        out[r] = np.roll(img[r], -int(shifts[r]))
    return out


def decode_field(data: bytes, dtype, h: int, w: int, px_offset) -> np.ndarray:
    """decode_field (png_tools.cpp:664-745): ZPNG as stored; PNG planes were destaggered by the writer."""
    z = decode_zpng_field(data, dtype, h, w)
    if z is not None:
        return z
    img = decode_png_field(data, dtype, h, w)
    return stagger(img, px_offset) if len(px_offset) else img


def decode_lidar_scan_msg(msg: bytes, h: int, w: int, px_offset) -> dict:
    t, _ = size_prefixed_root(msg, 0)
    types = t.vector(1, np.dtype([("field", "u1"), ("type", "u1")]))
    out = {"frame_id": t.scalar(5, "i"), "fields": {},
           "timestamp": t.vector(2, "<u8"), "measurement_id": t.vector(3, "<u2"), "status": t.vector(4, "<u4"),
           "packet_timestamp": t.vector(7, "<u8"), "pose": t.vector(6, "<f8"),
           "frame_status": t.scalar(9, "Q"), "alert_flags": t.vector(12, np.uint8)}
    for ch, ft in zip(t.table_vector(0), types if types is not None else []):
        data = ch.vector(0, np.uint8)
        name = CHAN_FIELD.get(int(ft["field"]), f"UNKNOWN{int(ft['field'])}")
        out["fields"][name] = decode_field(bytes(data) if data is not None else b"", FIELD_DTYPE[int(ft["type"])],
                                           h, w, px_offset)
    # custom_fields (fb_restore_fields, ouster_osf/src/fb_common.cpp:250-330; table Field of os_sensor/common.fbs:31-38):
    # name, tag, shape, class, data.  decode_field without pixel offsets (png_tools.cpp:667-706): a 1-D field is raw
    # bytes, anything else is an image of shape[0] x (size / shape[0]) -- ZPNG or PNG -- that is NOT staggered.
    out["custom_fields"] = {}
    for cf in t.table_vector(8):
        name = cf.string(0)
        tag = cf.scalar(1, "B")
        shape = [int(x) for x in (cf.vector(2, "<u8") if cf.vector(2, "<u8") is not None else [])]
        data = cf.vector(4, np.uint8)
        data = bytes(data) if data is not None else b""
        if tag not in CUSTOM_DTYPE or not shape:
            continue
        dt = np.dtype(CUSTOM_DTYPE[tag])
        if len(shape) == 1:
            arr = np.zeros(shape[0], dt)
            raw = np.frombuffer(data, np.uint8)
            arr.view(np.uint8)[:len(raw)] = raw[:arr.nbytes]
        else:
            rows = shape[0]
            cols = int(np.prod(shape)) // rows if rows else 0
            udt = np.dtype("<u%d" % dt.itemsize)
            arr = (decode_field(data, udt, rows, cols, []) if data and rows * cols else np.zeros((rows, cols), udt))
            arr = arr.view(dt).reshape(shape)
        out["custom_fields"][name] = {"array": arr, "field_class": int(cf.scalar(3, "q"))}
    return out
