"""GPU parity for SURVEY.md section 8 row f-4: OsfFrameDecoder (host inflate + ouster_hip_osf_unpack)
against the OSF oracle on the reference's own fixtures -- PNG (16-bit gray, RGBA, 8-bit gray) with the
stagger() back, ZPNG (1 / 2 / 4-byte pixels, colour planes + GB-RG) -- every plane and header bit-exact,
a whole file's messages in ONE launch, and the planes then feed the unchanged destagger / cartesian."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, has_gpu
from test_oracle_osf import LB, PNG8, ZPNG, _geometry

pytestmark = pytest.mark.gpu


def _sensor_info(core, meta):
    df = meta.get("lidar_data_format") or meta["data_format"]
    bi = meta.get("beam_intrinsics") or meta
    li = meta.get("lidar_intrinsics") or meta
    info = core.SensorInfo()
    fmt = core.DataFormat()
    fmt.pixels_per_column, fmt.columns_per_frame = df["pixels_per_column"], df["columns_per_frame"]
    fmt.columns_per_packet = df["columns_per_packet"]
    fmt.pixel_shift_by_row = list(df["pixel_shift_by_row"])
    fmt.udp_profile_lidar = core.UDPProfileLidar.from_string(df["udp_profile_lidar"])
    info.format = fmt
    info.beam_altitude_angles = list(bi["beam_altitude_angles"])
    info.beam_azimuth_angles = list(bi["beam_azimuth_angles"])
    b2l = np.eye(4)
    if "beam_to_lidar_transform" in bi:
        b2l = np.array(bi["beam_to_lidar_transform"], dtype=np.float64).reshape(4, 4)
    else:
        b2l[0, 3] = bi.get("lidar_origin_to_beam_origin_mm", 15.806)
    info.beam_to_lidar_transform = b2l
    info.lidar_to_sensor_transform = np.array(li["lidar_to_sensor_transform"], dtype=np.float64).reshape(4, 4)
    return info


@pytest.mark.parametrize("path", [LB, PNG8, ZPNG])
def test_osf_frames_match_oracle(oracle, path):
    from oracle import osf_oracle as Z
    from ouster_sdk_amd import core
    zf = Z.OsfFile(path)
    meta = list(zf.sensor_metadata().values())[0]
    h, w, shifts = _geometry(meta)
    pf = core.OsfFile(path)
    streams = pf.lidar_scan_streams()
    msgs = [m for (_, sid, m) in pf.messages() if sid in streams]
    assert msgs
    info = _sensor_info(core, meta)
    dec = core.OsfFrameDecoder(info)
    frames = dec.decode(msgs)                     # the whole file in one launch
    assert len(frames) == len(msgs)
    for fr, m in zip(frames, msgs):
        d = Z.decode_lidar_scan_msg(m, h, w, shifts)
        assert fr.frame_id == d["frame_id"]
        assert set(d["fields"]) == set(fr.fields)
        for name, want in d["fields"].items():
            got = fr.field(name)
            assert got.dtype == want.dtype and np.array_equal(got, want), name
        assert np.array_equal(fr.timestamp, d["timestamp"])
        assert np.array_equal(fr.status, d["status"])
        assert np.array_equal(fr.measurement_id, d["measurement_id"])
    # one message at a time gives the same frames (batching is transparent)
    single = dec.decode([msgs[-1]])[0]
    for name in Z.decode_lidar_scan_msg(msgs[-1], h, w, shifts)["fields"]:
        assert np.array_equal(single.field(name), frames[-1].field(name)), name
    # the planes feed the rest of the path unchanged: destagger + cartesian against the oracle
    O = oracle
    rng = frames[0].field("RANGE")
    assert np.array_equal(core.destagger(info, rng), O.destagger(rng, np.array(shifts, np.int32)))
    lut = core.XYZLut(info, False)
    xyz = lut(rng)
    d_, o_ = O.make_xyz_lut(w, h, 0.001, np.asarray(info.beam_to_lidar_transform), np.asarray(info.lidar_to_sensor_transform),
                            np.asarray(info.beam_azimuth_angles), np.asarray(info.beam_altitude_angles))
    want = O.cartesian(rng, d_, o_)
    assert np.abs(np.asarray(xyz).reshape(-1, 3) - want).max() <= 1e-9


@pytest.mark.parametrize("path", [LB, ZPNG])
def test_osf_device_batch_stays_in_hbm(oracle, path):
    """decode_device leaves [n][H][W] planes in HBM; destagger / cartesian run on them there and only the
    results are downloaded -- identical to the host-returning decode and to the oracle."""
    from oracle import osf_oracle as Z
    from ouster_sdk_amd import core
    O = oracle
    zf = Z.OsfFile(path)
    meta = list(zf.sensor_metadata().values())[0]
    h, w, shifts = _geometry(meta)
    pf = core.OsfFile(path)
    streams = pf.lidar_scan_streams()
    msgs = [m for (_, sid, m) in pf.messages() if sid in streams]
    info = _sensor_info(core, meta)
    dec = core.OsfFrameDecoder(info)
    frames = dec.decode(msgs)
    b = dec.decode_device(msgs)
    assert (b.n_frames, b.h, b.w) == (len(msgs), h, w)
    assert list(b.frame_ids) == [f.frame_id for f in frames]
    for name in b.fields():
        want = np.stack([f.field(name) for f in frames])
        got = np.empty_like(want)
        b.download(b.plane_ptr(name), got)
        assert np.array_equal(got, want), name
    rng = np.stack([f.field("RANGE") for f in frames])
    ds = np.empty_like(rng)
    b.download(b.destagger_ptr("RANGE"), ds)
    sh = np.array(shifts, np.int32)
    for i in range(len(frames)):
        assert np.array_equal(ds[i], O.destagger(rng[i], sh))
    lut = core.XYZLut(info, False)
    d_, o_ = O.make_xyz_lut(w, h, 0.001, np.asarray(info.beam_to_lidar_transform), np.asarray(info.lidar_to_sensor_transform),
                            np.asarray(info.beam_azimuth_angles), np.asarray(info.beam_altitude_angles))
    for f64, tol in ((True, 1e-9), (False, 1e-4)):
        xyz = np.empty((len(frames), h * w, 3), np.float64 if f64 else np.float32)
        b.download(b.cartesian_ptr(lut, f64), xyz)
        for i in range(len(frames)):
            assert np.abs(xyz[i].astype(np.float64) - O.cartesian(rng[i], d_, o_)).max() <= tol
    with pytest.raises(IndexError):
        b.plane_ptr("NOPE")


def test_zpng_fields_of_the_reference_codec(oracle):
    """ouster_hip_osf_unpack's ZPNG path (zstd on the host, prefix sums + colour transform on the GPU)
    against planes the REFERENCE's ZPNG_Compress produced: the committed vectors, and -- where
    oracle/_ref/libzpng_ref.so travelled with the snapshot -- freshly compressed full-size planes."""
    import zpng_ref
    from ouster_sdk_amd import core
    from test_oracle_osf import _zpng_vectors, _sha
    h, w, vec = _zpng_vectors()
    tag = {1: 1, 2: 2, 4: 3, 8: 4}   # itemsize -> ChanFieldType UINT8 / 16 / 32 / 64 (chanfield.h)

    def decoder(hh, ww):
        info = core.SensorInfo()
        fmt = core.DataFormat()
        fmt.pixels_per_column, fmt.columns_per_frame, fmt.columns_per_packet = hh, ww, 16
        fmt.pixel_shift_by_row = [0] * hh
        fmt.udp_profile_lidar = core.UDPProfileLidar.from_string("RNG19_RFL8_SIG16_NIR16")
        info.format = fmt
        return core.OsfFrameDecoder(info)

    names = list(vec)
    planes = decoder(h, w).decode_fields([(vec[n][1], int(tag[vec[n][0].itemsize])) for n in names])   # one launch
    for n, raw in zip(names, planes):
        dt, _, sha = vec[n]
        assert len(raw) == h * w * dt.itemsize and _sha(np.frombuffer(raw, dt)) == sha, n
    if not zpng_ref.available():
        return
    rng = np.random.default_rng(11)
    hh, ww = 128, 1024
    want, blobs = [], []
    for dt in (np.uint8, np.uint16, np.uint32, np.uint64):
        p = rng.integers(0, np.iinfo(dt).max, (hh, ww), dtype=np.uint64, endpoint=True).astype(dt)
        p[:, 1::2] = p[:, ::2] + 3                         # compressible: neighbouring columns correlate
        want.append(p)
        blobs.append((zpng_ref.compress(p), int(tag[np.dtype(dt).itemsize])))
    for p, raw in zip(want, decoder(hh, ww).decode_fields(blobs)):
        assert np.array_equal(np.frombuffer(raw, p.dtype).reshape(hh, ww), p), p.dtype


# ---------------------------------------------------------------------------------------------
# LidarScanMsg.custom_fields: every field whose name is not in the CHAN_FIELD enum
# ---------------------------------------------------------------------------------------------
CUSTOM1D = os.path.join(os.path.dirname(LB), "pose_delta_1_128.osf")


def test_osf_custom_fields_of_a_reference_file(oracle):
    """tests/osfs/pose_delta_1_128.osf stores six 1-D and four 2-D frame fields (IMU_TIMESTAMP, IMU_ACC, ..., the CHAR field POSITION_STRING) in custom_fields:
    the decoded frames carry them, equal to the oracle's restatement of fb_restore_fields (fb_common.cpp:250-330)."""
    from oracle import osf_oracle as Z
    from ouster_sdk_amd import core
    zf = Z.OsfFile(CUSTOM1D)
    meta = list(zf.sensor_metadata().values())[0]
    h, w, shifts = _geometry(meta)
    pf = core.OsfFile(CUSTOM1D)
    streams = pf.lidar_scan_streams()
    msgs = [m for (_, sid, m) in pf.messages() if sid in streams]
    frames = core.OsfFrameDecoder(_sensor_info(core, meta)).decode(msgs)
    assert len(frames) == len(msgs) == 2
    for fr, m in zip(frames, msgs):
        d = Z.decode_lidar_scan_msg(m, h, w, shifts)
        assert set(d["custom_fields"]) == {"POSITION_TIMESTAMP", "IMU_STATUS", "IMU_PACKET_TIMESTAMP", "IMU_ALERT_FLAGS",
                                           "IMU_MEASUREMENT_ID", "IMU_TIMESTAMP", "IMU_ACC", "IMU_GYRO", "POSITION_LAT_LONG",
                                           "POSITION_STRING"}
        assert d["custom_fields"]["POSITION_STRING"]["array"].dtype == np.dtype("S1")   # a CHAR field: bytes through the 8-bit codec
        assert d["custom_fields"]["IMU_ACC"]["array"].ndim == 2       # 2-D float fields: through the image codec
        assert set(fr.fields) == set(d["fields"]) | set(d["custom_fields"])
        for name, c in d["custom_fields"].items():
            got = fr.field(name)
            assert got.dtype == c["array"].dtype and got.shape == c["array"].shape and np.array_equal(got, c["array"]), name
        for name, want in d["fields"].items():
            assert np.array_equal(fr.field(name), want), name
    assert any(np.asarray(frames[0].field("IMU_TIMESTAMP")).any() for _ in (0,))


def _fb_message(frame_id, custom):
    """A size-prefixed LidarScanMsg flatbuffer with only frame_id (field 5) and custom_fields (field 8), assembled by hand
    front to back (every offset points forward, which is all a reader needs).  custom: [(name, tag, shape, class, data)]."""
    import struct
    buf = bytearray(8)                                        # size prefix + root offset, patched at the end

    def table(n_fields, inline):                              # inline: {field index: (struct fmt, value)}; returns (pos, {idx: field pos})
        offs, body = {}, bytearray(4)                         # soffset placeholder
        for idx, (fmt, val) in sorted(inline.items()):
            while len(body) % struct.calcsize(fmt):
                body += b"\0"
            offs[idx] = len(body)
            body += struct.pack("<" + fmt, val)
        vt = struct.pack("<HH", 4 + 2 * n_fields, len(body)) + b"".join(struct.pack("<H", offs.get(i, 0)) for i in range(n_fields))
        while (len(buf) + len(vt)) % 8:
            buf.append(0)
        vt_pos = len(buf)
        buf.extend(vt)
        pos = len(buf)
        struct.pack_into("<i", body, 0, pos - vt_pos)
        buf.extend(body)
        return pos, {i: pos + o for i, o in offs.items()}

    def point(field_pos, target):
        struct.pack_into("<I", buf, field_pos, target - field_pos)

    def blob(data, elem=1):                                   # [u32 count][bytes]
        while len(buf) % 8:
            buf.append(0)
        pos = len(buf)
        buf.extend(struct.pack("<I", len(data) // elem) + bytes(data))
        return pos

    root, rf = table(13, {5: ("i", frame_id), 8: ("I", 0)})
    struct.pack_into("<I", buf, 4, root - 4)
    while len(buf) % 4:
        buf.append(0)
    vec = len(buf)
    buf.extend(struct.pack("<I", len(custom)) + b"\0" * (4 * len(custom)))
    point(rf[8], vec)
    for i, (name, tag, shape, cls, data) in enumerate(custom):
        t, tf = table(6, {0: ("I", 0), 1: ("B", tag), 2: ("I", 0), 3: ("q", cls), 4: ("I", 0), 5: ("Q", len(data))})
        point(vec + 4 + 4 * i, t)
        point(tf[0], blob(name.encode() + b"\0")); struct.pack_into("<I", buf, len(buf) - len(name) - 1 - 4, len(name))
        point(tf[2], blob(struct.pack("<%dQ" % len(shape), *shape), 8))
        point(tf[4], blob(data))
    struct.pack_into("<I", buf, 0, len(buf) - 4)
    return bytes(buf)


def _png_gray16(img):
    """A minimal 16-bit gray PNG (filter 0 on every row, big-endian samples) of a 2-D array."""
    import struct
    import zlib

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    h, w = img.shape
    raw = b"".join(b"\0" + img[r].astype(">u2").tobytes() for r in range(h))
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def test_osf_custom_fields_2d_and_1d_from_a_hand_built_message(oracle):
    """A message with a 2-D u16 pixel field (PNG, NOT staggered back: custom fields are stored as they are), a
    (w, 3)-shaped u16 column field collapsed to w x 3 by the writer, and a 1-D u32 frame field of raw bytes."""
    from oracle import osf_oracle as Z
    from ouster_sdk_amd import core
    zf = Z.OsfFile(LB)
    meta = list(zf.sensor_metadata().values())[0]
    h, w, shifts = _geometry(meta)
    g = np.random.default_rng(5)
    px = g.integers(0, 65536, (h, w)).astype(np.uint16)
    colf = g.integers(0, 65536, (w, 3)).astype(np.uint16)
    one = g.integers(0, 2 ** 32, 7).astype(np.uint32)
    msg = _fb_message(4321, [("MY_PIXELS", 2, [h, w], 1, _png_gray16(px)),
                             ("MY_COLUMNS", 2, [w, 3], 2, _png_gray16(colf)),
                             ("MY_FRAME_VALUES", 3, [7], 4, one.tobytes())])
    d = Z.decode_lidar_scan_msg(msg, h, w, shifts)           # the oracle reads the same bytes
    assert d["frame_id"] == 4321 and np.array_equal(d["custom_fields"]["MY_PIXELS"]["array"], px)
    assert np.array_equal(d["custom_fields"]["MY_COLUMNS"]["array"], colf)
    fr = core.OsfFrameDecoder(_sensor_info(core, meta)).decode([msg])[0]
    assert fr.frame_id == 4321 and set(fr.fields) == {"MY_PIXELS", "MY_COLUMNS", "MY_FRAME_VALUES"}
    assert np.array_equal(fr.field("MY_PIXELS"), px) and fr.field("MY_PIXELS").dtype == np.uint16
    assert fr.field("MY_COLUMNS").shape == (w, 3) and np.array_equal(fr.field("MY_COLUMNS"), colf)
    assert np.array_equal(fr.field("MY_FRAME_VALUES"), one)


# ---------------------------------------------------------------------------------------------
# PNG scanline filters on the GPU (round 5, k_osf_png_unfilter): every filter type, every pixel size
# ---------------------------------------------------------------------------------------------
def _png_with_filters(pixels: np.ndarray, depth: int, colour: int, filters) -> bytes:
    """A PNG whose row y is filtered with type filters[y] (PNG specification section 9): the encoder side of what the decoder
    must reverse; pixels [h, w * bpp] uint8."""
    import struct
    import zlib
    h, stride = pixels.shape
    bpp = {0: 1, 2: 3, 6: 4}[colour] * depth // 8
    raw = bytearray()
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        cur = pixels[y].astype(np.int32)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        ft = int(filters[y])
        if ft == 0:
            pred = np.zeros(stride, np.int32)
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (left + prev) >> 1
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        raw.append(ft)
        raw += ((cur - pred) & 0xFF).astype(np.uint8).tobytes()
        prev = cur

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)
    ihdr = struct.pack(">IIBBBBB", stride // bpp, h, depth, colour, 0, 0, 0)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b"")


@pytest.mark.parametrize("h,w", [(128, 1024), (64, 512), (70, 130), (200, 96), (1, 64), (5, 3)])
def test_png_filters_on_the_gpu_equal_the_host_and_the_oracle(oracle, h, w):
    """Every scanline filter (None / Sub / Up / Average / Paeth, and rows that mix them in every order), every pixel size the
    OSF writer produces (8 / 16-bit gray, RGB8, RGBA8, RGBA16): OsfFrameDecoder.decode_fields with the filters reversed on the
    GPU == with the filters reversed on the host == osf_oracle.decode_field (pinned on the reference's own .osf files).  Image
    heights that are not a multiple of the kernel's 64-row band, one-row and three-column images."""
    from oracle import osf_oracle as OO
    from ouster_sdk_amd import core
    rng = np.random.default_rng(h * 1000 + w)
    info = core.SensorInfo()
    fmt = core.DataFormat()
    fmt.pixels_per_column, fmt.columns_per_frame, fmt.columns_per_packet = h, w, 1
    # stagger() is not the subject here: random shifts where np.roll is the reference's arithmetic (power-of-two widths), none
    # elsewhere (the reference's size_t expression is not a roll there: tests/test_oracle_ref_core.py)
    pow2 = w & (w - 1) == 0
    fmt.pixel_shift_by_row = [int(x) if pow2 else 0 for x in rng.integers(-w, w, h)]
    fmt.udp_profile_lidar = core.UDPProfileLidar.from_string("RNG19_RFL8_SIG16_NIR16")
    info.format = fmt
    kinds = [(8, 0, np.uint8, 1), (16, 0, np.uint16, 2), (8, 2, np.uint32, 3), (8, 6, np.uint32, 3), (16, 6, np.uint64, 4)]
    blobs, want = [], []
    for depth, colour, dt, tag in kinds:
        bpp = {0: 1, 2: 3, 6: 4}[colour] * depth // 8
        for mode in ("all0", "all1", "all2", "all3", "all4", "mixed"):
            px = rng.integers(0, 256, (h, w * bpp), dtype=np.uint8)
            px[:, bpp:] = (px[:, bpp:] // 8 + px[:, :-bpp]).astype(np.uint8)     # neighbouring pixels correlate
            filters = rng.integers(0, 5, h) if mode == "mixed" else np.full(h, int(mode[-1]))
            blob = _png_with_filters(px, depth, colour, filters)
            blobs.append((blob, tag))
            want.append(OO.decode_field(blob, dt, h, w, fmt.pixel_shift_by_row))
            if mode == "mixed":   # the oracle's unfilter gives the pixels back (the test's encoder and the oracle agree)
                assert np.array_equal(OO.png_pixels(blob)[0], px)
    got = {}
    for on in (True, False):
        dec = core.OsfFrameDecoder(info)
        dec.device_unfilter = on
        assert dec.device_unfilter == on
        got[on] = dec.decode_fields(blobs)
    for i, ((depth, colour, dt, tag), w_) in enumerate(zip([k for k in kinds for _ in range(6)], want)):
        a = np.frombuffer(got[True][i], dt).reshape(h, w)
        b = np.frombuffer(got[False][i], dt).reshape(h, w)
        assert np.array_equal(a, w_), ("gpu unfilter", i, depth, colour)
        assert np.array_equal(b, w_), ("host unfilter", i, depth, colour)
    # what libpng refuses is refused: a filter type above 4
    bad = bytearray(_png_with_filters(rng.integers(0, 256, (h, w), dtype=np.uint8), 8, 0, np.zeros(h, int)))
    import struct
    import zlib
    raw = bytearray(zlib.decompress(bytes(OO_idat(bad))))
    raw[0] = 7
    blob = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + _chunk(b"IDAT", zlib.compress(bytes(raw))) + _chunk(b"IEND", b"")
    for on in (True, False):
        dec = core.OsfFrameDecoder(info)
        dec.device_unfilter = on
        with pytest.raises(RuntimeError, match="could not decode field"):
            dec.decode_fields([(blob, 1)])
        # ... also in the middle of a batch staged by the decoder's crew of threads, which then serves the next call: a stream
        # cut short, one split over several IDAT chunks, and the decoder's staging buffers growing between calls
        good = blobs[:7]
        cut = _rechunk(blobs[0][0], lambda z: [z[:len(z) // 2]])
        with pytest.raises(RuntimeError, match="could not decode field"):
            dec.decode_fields(good + [(blob, 1)] + good)
        with pytest.raises(RuntimeError, match="could not decode field"):
            dec.decode_fields(good + [(cut, blobs[0][1])])
        split = _rechunk(blobs[5][0], lambda z: [z[:1], z[1:len(z) // 3], z[len(z) // 3:]])
        again = dec.decode_fields([(split, blobs[5][1])] + blobs * 2)
        assert bytes(again[0]) == bytes(got[on][5])
        for i in range(len(blobs)):
            assert bytes(again[1 + i]) == bytes(got[on][i]) and bytes(again[1 + len(blobs) + i]) == bytes(got[on][i])


def _rechunk(png, parts):
    """The same PNG with its zlib stream cut into the IDAT chunks parts(stream) returns."""
    import struct
    z = bytes(OO_idat(bytearray(png)))
    ihdr = png[8:8 + 25]
    return png[:8] + ihdr + b"".join(_chunk(b"IDAT", p) for p in parts(z)) + _chunk(b"IEND", b"")


def _chunk(typ, body):
    import struct
    import zlib
    return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)


def OO_idat(png: bytes) -> bytes:
    import struct
    pos, out = 8, b""
    while pos + 8 <= len(png):
        n, typ = struct.unpack_from(">I4s", png, pos)
        if typ == b"IDAT":
            out += bytes(png[pos + 8:pos + 8 + n])
        pos += 12 + n
    return out
