"""CPU-only: SURVEY.md section 8 row f-4 (OSF field planes).

  * the OSF oracle (oracle/osf_oracle.py, a Python restatement of the reference's container walk and
    decode_field) is PINNED: the planes and column headers it decodes from the reference's
    tests/osfs/OS-1-128_v2.3.0_1024x10_lb_n3.osf equal the frames the packet oracle batches from the capture
    the reference wrote that file from (sha256 goldens made by tests/golden/make_golden.py);
  * the product's host half (C++ OsfFile, LidarScanMsgView, stage_field: flatbuffers, CRC32, zlib +
    PNG filters, zstd) equals the oracle's, block for block and byte for byte, on the PNG and the ZPNG
    fixtures -- no GPU involved;
  * corrupted files fail loudly like the reference's reader.
ZPNG: the reference's tests hold no golden for a ZPNG plane, but its codec is one self-contained file:
oracle/Makefile compiles /root/reference/thirdparty/zpng/zpng.cpp into oracle/_ref/libzpng_ref.so, and the
restatement in osf_oracle.py is pinned on it three ways -- the reference decoder on the reference's own
ZPNG fixture, reference-compressed random planes of every layout, and the committed vectors of
tests/golden/osf/zpng_ref_vectors.json (made by tests/golden/make_zpng_golden.py) where the library is absent.
"""
import hashlib
import json
import os
import shutil

import numpy as np
import pytest

from conftest import GOLDEN

OSF_DIR = os.path.join(GOLDEN, "osf")
LB = os.path.join(OSF_DIR, "OS-1-128_v2.3.0_1024x10_lb_n3.osf")
PNG8 = os.path.join(OSF_DIR, "OS-0-128_v3.0.1_1024x10_20241017_141645.osf")
ZPNG = os.path.join(OSF_DIR, "single_scan_016.osf")


def _geometry(meta):
    df = meta.get("lidar_data_format") or meta["data_format"]
    return df["pixels_per_column"], df["columns_per_frame"], df["pixel_shift_by_row"]


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_osf_oracle_is_pinned_on_the_capture_the_file_was_written_from():
    from oracle import osf_oracle as Z
    want = json.load(open(os.path.join(OSF_DIR, "lb_n3_pcap_planes.json")))["frames"]
    f = Z.OsfFile(LB)
    assert f.header_ok and f.metadata_ok and f.status == 2
    h, w, shifts = _geometry(list(f.sensor_metadata().values())[0])
    msgs = f.messages()
    assert len(msgs) == 3 == len(want)
    for ts, sid, m in msgs:
        d = Z.decode_lidar_scan_msg(m, h, w, shifts)
        g = want[str(d["frame_id"])]
        for name in ("RANGE", "REFLECTIVITY", "NEAR_IR"):
            assert _sha(d["fields"][name].astype(np.uint64)) == g[name], (d["frame_id"], name)
        assert _sha(d["timestamp"]) == g["timestamp"] and _sha(d["status"]) == g["status"]
        assert _sha(d["measurement_id"]) == g["measurement_id"]


@pytest.mark.parametrize("path", [LB, PNG8, ZPNG])
def test_host_half_matches_the_oracle(path):
    """C++ container walk + entropy decoding vs the oracle: same blocks, same staged pixel bytes."""
    import struct
    from oracle import osf_oracle as Z
    from ouster_sdk_amd import core
    zf, pf = Z.OsfFile(path), core.OsfFile(path)
    assert pf.version == zf.version and pf.id == zf.id
    assert pf.metadata_types() == {k: v[0] for k, v in zf.entries.items()}
    assert pf.lidar_scan_streams() == zf.lidar_streams()
    assert {k: json.loads(v) for k, v in pf.sensor_metadata_json().items()} == zf.sensor_metadata()
    zm, pm = zf.messages(), pf.messages()
    assert [(a, b) for a, b, _ in zm] == [(a, b) for a, b, _ in pm]
    h, w, _ = _geometry(list(zf.sensor_metadata().values())[0])
    streams = zf.lidar_streams()
    n_checked = 0
    for (ts, sid, zbytes), (_, _, pbytes) in zip(zm, pm):
        assert zbytes == pbytes
        if sid not in streams:
            continue
        t, _ = Z.size_prefixed_root(zbytes, 0)
        chans = t.table_vector(0)
        staged = core.osf_stage_fields(pbytes, h, w)
        assert len(staged) == len(chans)
        for ch, (name, typ, enc, pb, data) in zip(chans, staged):
            raw = bytes(ch.vector(0, np.uint8))
            if raw[:2] == b"\xf8\xfb":
                _, zw, zh, c, bpc = struct.unpack_from("<HHHBB", raw, 0)
                assert enc == 6 and pb == c * bpc and (zw, zh) == (w, h)
                assert data == Z.zstd_decompress(raw[8:], w * h * pb)
            else:
                px, pw, ph, depth, colour = Z.png_pixels(raw)
                assert (pw, ph) == (w, h) and pb == px.shape[1] // w
                assert enc == {(0, 8): 1, (0, 16): 2, (2, 8): 3, (6, 8): 4, (6, 16): 5}[(colour, depth)]
                assert data == px.tobytes()
            n_checked += 1
    assert n_checked >= 3


def test_zpng_oracle_self_consistency():
    """Independent of the reference library: check the restated codec against its own inverse
    (left-delta + GB-RG filter of zpng.cpp:69-100, 243-297 re-applied to the decoded plane reproduces the
    residuals) and the decoded values against the masks the wire format allows."""
    from oracle import osf_oracle as Z
    f = Z.OsfFile(ZPNG)
    h, w, shifts = _geometry(list(f.sensor_metadata().values())[0])
    ts, sid, m = f.messages()[0]
    d = Z.decode_lidar_scan_msg(m, h, w, shifts)
    assert set(d["fields"]) >= {"RANGE", "RANGE2", "REFLECTIVITY", "NEAR_IR"}
    assert int(d["fields"]["RANGE"].max()) < (1 << 20) and np.count_nonzero(d["fields"]["RANGE"]) > 1000
    t, _ = Z.size_prefixed_root(m, 0)
    for ch, name in zip(t.table_vector(0), d["fields"]):
        raw = bytes(ch.vector(0, np.uint8))
        plane = d["fields"][name]
        pb = plane.dtype.itemsize
        b = plane.view(np.uint8).reshape(h, w, pb).astype(np.int32)
        delta = b.copy()
        delta[:, 1:] -= b[:, :-1]
        delta &= 0xFF
        if pb == 4:   # forward GB-RG: y = B, u = G - B, v = G - R
            r, g, bl, a = (delta[..., k] for k in range(4))
            resid = np.stack([bl, (g - bl) & 0xFF, (g - r) & 0xFF, a]).astype(np.uint8).tobytes()
        else:
            resid = delta.astype(np.uint8).tobytes()
        assert resid == Z.zstd_decompress(raw[8:], w * h * pb), name


def test_corrupted_files_fail_loudly(tmp_path):
    from ouster_sdk_amd import core
    data = bytearray(open(ZPNG, "rb").read())
    bad = tmp_path / "bad_crc.osf"
    flipped = bytearray(data)
    flipped[len(data) // 2] ^= 0x5A                      # inside the chunk: its CRC32 no longer matches
    bad.write_bytes(flipped)
    f = core.OsfFile(str(bad))
    with pytest.raises(RuntimeError, match="chunk crc32 mismatch"):
        f.messages()
    notosf = tmp_path / "x.osf"
    notosf.write_bytes(b"\x00" * 64)
    with pytest.raises(RuntimeError, match="not an OSF file"):
        core.OsfFile(str(notosf))
    with pytest.raises(RuntimeError, match="cannot open"):
        core.OsfFile(str(tmp_path / "missing.osf"))
    # a field whose bytes are neither PNG nor ZPNG (the reference's bad_encoding.osf case)
    from oracle import osf_oracle as Z
    h, w, _ = _geometry(list(Z.OsfFile(ZPNG).sensor_metadata().values())[0])
    good = core.OsfFile(ZPNG).messages()[0][2]
    assert len(core.osf_stage_fields(good, h, w)) == 9
    broken = good.replace(b"\xf8\xfb" + bytes([w & 255, w >> 8, h & 255, h >> 8]),
                          b"\x00\x50" + bytes([w & 255, w >> 8, h & 255, h >> 8]))
    assert broken != good
    with pytest.raises(RuntimeError, match="could not decode field"):
        core.osf_stage_fields(broken, h, w)
    with pytest.raises(RuntimeError, match="Invalid allocation"):   # ZPNG image of another size
        core.osf_stage_fields(good, h, w // 2)


def _zpng_vectors():
    import base64
    d = json.load(open(os.path.join(OSF_DIR, "zpng_ref_vectors.json")))
    return d["h"], d["w"], {k: (np.dtype(v["dtype"]), base64.b64decode(v["zpng"]), v["sha256"])
                            for k, v in d["vectors"].items()}


def test_zpng_oracle_decodes_the_reference_codecs_vectors():
    """Committed vectors: planes compressed by the reference's ZPNG_Compress in the layouts its OSF writer
    uses (u8: 1x1 B, u16: 1x2 B, u32: 4x1 B with the colour transform, u64: 4x2 B)."""
    from oracle import osf_oracle as Z
    h, w, vec = _zpng_vectors()
    assert len(vec) >= 10
    for name, (dt, blob, sha) in vec.items():
        got = Z.decode_zpng_field(blob, dt, h, w)
        assert got is not None and got.dtype == dt and _sha(got) == sha, name


def test_zpng_oracle_matches_the_reference_decoder():
    """oracle/_ref (the reference's zpng.cpp compiled as it lies): its decoder and the restatement agree on
    every ZPNG field of the reference's own fixture, and on reference-compressed random planes."""
    import zpng_ref
    if not zpng_ref.available():
        pytest.skip("oracle/_ref/libzpng_ref.so not built (no /root/reference at build time)")
    from oracle import osf_oracle as Z
    f = Z.OsfFile(ZPNG)
    h, w, shifts = _geometry(list(f.sensor_metadata().values())[0])
    n = 0
    for ts, sid, m in f.messages():
        if sid not in f.lidar_streams():
            continue
        d = Z.decode_lidar_scan_msg(m, h, w, shifts)
        t, _ = Z.size_prefixed_root(m, 0)
        for ch, name in zip(t.table_vector(0), d["fields"]):
            raw = bytes(ch.vector(0, np.uint8))
            if raw[:2] != b"\xf8\xfb":
                continue
            px, zw, zh, c, bpc = zpng_ref.decompress(raw)
            plane = d["fields"][name]
            assert (zw, zh) == (w, h) and c * bpc == plane.dtype.itemsize
            assert px == plane.tobytes(), name            # ZPNG planes are stored staggered, as they are
            n += 1
    assert n >= 4
    rng = np.random.default_rng(7)
    for dt in (np.uint8, np.uint16, np.uint32, np.uint64):
        for hh, ww in ((16, 48), (5, 7), (64, 128)):
            p = rng.integers(0, np.iinfo(dt).max, (hh, ww), dtype=np.uint64, endpoint=True).astype(dt)
            p[:, ::3] >>= 3                                   # mixed entropy
            blob = zpng_ref.compress(p)
            assert np.array_equal(Z.decode_zpng_field(blob, np.dtype(dt), hh, ww), p), (dt, hh, ww)


_FUZZ = r"""
import os, sys, random
sys.path.insert(0, sys.argv[1])
from ouster_sdk_amd import core
src, tmp, seed, n = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
data = open(src, "rb").read()
rnd = random.Random(seed)
ok = bad = 0
for it in range(n):
    b = bytearray(data)
    kind = rnd.randrange(4)
    if kind == 0:                                   # flip a few bytes anywhere
        for _ in range(rnd.randrange(1, 8)):
            b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
    elif kind == 1:                                 # overwrite a dword with an extreme value
        p = rnd.randrange(len(b) - 4)
        b[p:p + 4] = rnd.choice([b"\xff\xff\xff\xff", b"\x00\x00\x00\x00", b"\xff\xff\xff\x7f", b"\x00\x00\x00\x80"])
    elif kind == 2:                                 # truncate
        del b[rnd.randrange(16, len(b)):]
    else:                                           # garbage run
        p = rnd.randrange(len(b) - 64)
        b[p:p + 64] = bytes(rnd.randrange(256) for _ in range(64))
    path = os.path.join(tmp, "f.osf")
    open(path, "wb").write(bytes(b))
    try:
        f = core.OsfFile(path)
        streams = f.lidar_scan_streams()
        f.sensor_metadata_json()
        for ts, sid, m in f.messages():
            if sid in streams:
                core.osf_stage_fields(m, int(sys.argv[6]), int(sys.argv[7]))
        ok += 1
    except (RuntimeError, ValueError, IndexError, OverflowError, MemoryError):
        bad += 1
# the same on single messages, behind the CRCs: flatbuffer offsets, vector lengths, PNG chunks, zlib and
# zstd streams are all reachable here
f = core.OsfFile(src)
streams = f.lidar_scan_streams()
msgs = [m for ts, sid, m in f.messages() if sid in streams]
mok = mbad = 0
for it in range(n * 4):
    b = bytearray(msgs[it % len(msgs)])
    kind = rnd.randrange(4)
    if kind == 0:
        for _ in range(rnd.randrange(1, 6)):
            b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
    elif kind == 1:
        p = rnd.randrange(min(len(b) - 4, 4096) if rnd.random() < 0.7 else len(b) - 4)   # mostly the tables up front
        b[p:p + 4] = rnd.choice([b"\xff\xff\xff\xff", b"\x00\x00\x00\x00", b"\xff\xff\xff\x7f", b"\x00\x00\x00\x80", b"\x10\x00\x00\x00"])
    elif kind == 2:
        del b[rnd.randrange(8, len(b)):]
    else:
        p = rnd.randrange(len(b) - 32)
        b[p:p + 32] = bytes(rnd.randrange(256) for _ in range(32))
    try:
        core.osf_stage_fields(bytes(b), int(sys.argv[6]), int(sys.argv[7]))
        mok += 1
    except (RuntimeError, ValueError, IndexError, OverflowError, MemoryError):
        mbad += 1
print("FUZZ_DONE", ok, bad, mok, mbad)
"""


@pytest.mark.parametrize("path", [LB, ZPNG])
def test_host_half_survives_corrupted_files(path, tmp_path):
    """Bit flips, extreme length words, truncation and garbage runs anywhere in a file: the container walk,
    the flatbuffer reads, CRC checks, zlib / PNG unfilter and zstd either succeed or raise -- the process
    must not crash (run in a child so that a segfault is a test failure, not a dead test run)."""
    import subprocess
    import sys
    from oracle import osf_oracle as Z
    h, w, _ = _geometry(list(Z.OsfFile(path).sensor_metadata().values())[0])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FUZZ, root, path, str(tmp_path), "1234", "150", str(h), str(w)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FUZZ_DONE" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    ok, bad, mok, mbad = map(int, r.stdout.split("FUZZ_DONE")[1].split()[:4])
    assert ok + bad == 150 and bad > 20      # the CRCs catch most corruptions; a few land in slack bytes
    assert mok + mbad == 600 and mbad > 50   # behind the CRCs: decoders and bounds checks do the rejecting
