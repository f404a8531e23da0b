"""Pins the CPU oracle against the reference's own golden vectors (CPU only).

Goldens (copied into tests/golden by tests/golden/make_golden.py):
  * md5 digests per packet stream and first frame -- reference
    tests/pcaps/<capture>_digest.json, checked the way
    python/src/ouster/sdk/core/_digest.py:55-82 computes them
  * per-field hash snapshots -- tests/frame_batcher_test.cpp:553-610
  * first-packet header known answers -- tests/parsing_benchmark_test.cpp:86-115
  * field bit widths -- tests/packet_format_test.cpp:63-151
  * frame_id_difference KATs -- tests/packet_format_test.cpp:776-820
  * stored CRC64 == computed -- python/tests/test_parsing.py:84-98
"""
import collections
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, PCAPS

DIGEST_CAPTURES = [
    "OS-0-128-U1_v2.3.0_1024x10",
    "OS-0-32-U1_v2.2.0_1024x10",
    "OS-2-128-U1_v2.3.0_1024x10",
    "OS-2-32-U0_v2.0.0_1024x10",
    "OS-1-32-G_v2.1.1_1024x10",
]


def _load(O, base):
    cal = O.calib_from_json(os.path.join(PCAPS, base + ".json"))
    pf = cal.packet_format()
    pk = O.lidar_packets_from_pcap(os.path.join(PCAPS, base + ".pcap"), pf)
    return cal, pf, pk


def _md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("base", DIGEST_CAPTURES)
def test_packet_digest(oracle, base):
    O = oracle
    cal, pf, pk = _load(O, base)
    dig = json.load(open(os.path.join(PCAPS, base + "_digest.json")))
    hs = collections.defaultdict(hashlib.md5)
    for p in pk:
        for h in ("TIMESTAMP", "MEASUREMENT_ID", "STATUS", "FRAME_ID"):
            hs[h].update(O.packet_header(pf, h, p).tobytes())
        for n in pf.field_names():
            hs[n].update(O.packet_field(pf, n, p).tobytes())
    checked = 0
    for k, v in dig["packet_hash"].items():
        if k == "ENCODER_COUNT":  # skipped by the reference too (_digest.py:47-50)
            continue
        assert hs[k].hexdigest() == v, k
        checked += 1
    assert checked >= 8


@pytest.mark.parametrize("base", DIGEST_CAPTURES)
@pytest.mark.parametrize("force_col", [False, True])
def test_frame_digest(oracle, base, force_col):
    O = oracle
    cal, pf, pk = _load(O, base)
    dig = json.load(open(os.path.join(PCAPS, base + "_digest.json")))["scans"][0]
    fr = O.Frame.for_profile(cal.profile, cal.h, cal.w, cal.cpp, with_window=False)
    b = O.Batcher(pf, init_id=cal.init_id)
    if force_col:
        b.force_col_path()
    done = [b.batch(p, 1234, fr) for p in pk]
    assert done.index(True) == 63 and sum(done) == 1
    res = {"FRAME_ID": str(fr.frame_id),
           "TIMESTAMP": _md5(fr.timestamp.astype(np.uint64)),
           "STATUS": _md5(fr.status.astype(np.uint64)),
           "MEASUREMENT_ID": _md5(fr.measurement_id.astype(np.uint16))}
    for n in fr.plane_names():
        res[n] = _md5(fr.plane(n))
    for k, v in dig.items():
        if k == "ENCODER_COUNT":
            continue
        assert res[k] == v, k


def _matrix_hash(a):
    """tests/frame_batcher_test.cpp:600-610 with libstdc++'s identity std::hash."""
    seed = 0
    M = (1 << 64) - 1
    for e in a.reshape(-1).tolist():
        seed ^= (e + 0x9E3779B9 + ((seed << 6) & M) + (seed >> 2)) & M
    return seed


def test_snapshot_hashes(oracle):
    O = oracle
    snaps = json.load(open(os.path.join(GOLDEN, "snapshot_hashes.json")))
    assert len(snaps) == 5
    for base, fields in snaps.items():
        cal, pf, pk = _load(O, base)
        fr = O.Frame.for_profile(cal.profile, cal.h, cal.w, cal.cpp, with_window=False)
        b = O.Batcher(pf, init_id=cal.init_id)
        for p in pk:
            b.batch(p, 1234, fr)
        for name, want in fields.items():
            assert _matrix_hash(fr.plane(name)) == want, (base, name)


# tests/parsing_benchmark_test.cpp:86-115 (frame_id, init_id, prod_sn, col 7 timestamp... )
KNOWN_HEADERS = {
    "OS-0-128-U1_v2.3.0_1024x10": (1491, 5431292, 122150000150, 1462560143810),
    "OS-0-32-U1_v2.2.0_1024x10": (1453, 9599938, 992137000142, 515817575400),
    "OS-1-128_767798045_1024x10_20230712_120049": (229, 390076, 122246000293, 647840675576),
    "OS-2-128-U1_v2.3.0_1024x10": (1259, 5431293, 992210000957, 765697732720),
}


@pytest.mark.parametrize("base", sorted(KNOWN_HEADERS))
def test_known_headers(oracle, base):
    O = oracle
    cal, pf, pk = _load(O, base)
    L = O.lib()
    p0 = pk[0]
    fid, init, sn, ts = KNOWN_HEADERS[base]
    assert L.ora_frame_id(C.byref(pf), p0.ctypes.data) == fid
    assert L.ora_init_id(C.byref(pf), p0.ctypes.data) == init
    assert L.ora_prod_sn(C.byref(pf), p0.ctypes.data) == sn
    assert L.ora_packet_type(C.byref(pf), p0.ctypes.data) == 1
    assert int(O.packet_header(pf, "TIMESTAMP", p0)[7]) == ts
    assert int(O.packet_header(pf, "STATUS", p0)[7]) == 1
    assert int(O.packet_header(pf, "MEASUREMENT_ID", p0)[7]) == 7


BITNESS = {  # tests/packet_format_test.cpp:71-121
    "LEGACY": {"RANGE": 20, "FLAGS": 4, "REFLECTIVITY": 8, "SIGNAL": 16, "NEAR_IR": 16,
               "RAW32_WORD1": 32, "RAW32_WORD2": 32, "RAW32_WORD3": 32},
    "RNG15_RFL8_NIR8": {"RANGE": 15, "FLAGS": 1, "REFLECTIVITY": 8, "NEAR_IR": 8,
                        "RAW32_WORD1": 32},
    "RNG19_RFL8_SIG16_NIR16": {"RANGE": 19, "FLAGS": 5, "REFLECTIVITY": 8, "SIGNAL": 16,
                               "NEAR_IR": 16, "WINDOW": 8, "RAW32_WORD1": 32,
                               "RAW32_WORD2": 32, "RAW32_WORD3": 32},
    "RNG19_RFL8_SIG16_NIR16_DUAL": {"RANGE": 19, "FLAGS": 5, "REFLECTIVITY": 8, "RANGE2": 19,
                                    "FLAGS2": 5, "REFLECTIVITY2": 8, "SIGNAL": 16,
                                    "SIGNAL2": 16, "NEAR_IR": 16, "WINDOW": 8,
                                    "RAW32_WORD1": 32, "RAW32_WORD2": 32, "RAW32_WORD3": 32,
                                    "RAW32_WORD4": 32},
    "FUSA_RNG15_RFL8_NIR8_DUAL": {"RANGE": 15, "FLAGS": 1, "REFLECTIVITY": 8, "RANGE2": 15,
                                  "FLAGS2": 1, "REFLECTIVITY2": 8, "NEAR_IR": 8, "WINDOW": 8,
                                  "RAW32_WORD1": 32, "RAW32_WORD2": 32},
}


@pytest.mark.parametrize("profile", sorted(BITNESS))
def test_field_bitness(oracle, profile):
    O = oracle
    pf = O.packet_format(profile, 128, 16, 1024)
    assert sorted(pf.field_names()) == sorted(BITNESS[profile])
    for name, bits in BITNESS[profile].items():
        f = pf.field(name)
        vm = O.lib().ora_value_mask(C.byref(f))
        assert bin(vm).count("1") == bits, name
        if f.shift < 0:
            assert vm == f.mask << -f.shift
        else:
            assert vm == f.mask >> f.shift


@pytest.mark.parametrize("header_type,max_id", [(0, 0xFFFF), (1, 0xFFFFFFFF)])
def test_frame_id_difference(oracle, header_type, max_id):
    O = oracle
    pf = O.packet_format("RNG19_RFL8_SIG16_NIR16", 128, 16, 1024, header_type)
    assert pf.max_frame_id == max_id
    d = lambda a, b: O.lib().ora_frame_id_difference(C.byref(pf), a, b)
    assert d(0, 0) == 0
    assert d(0, 1) == 1
    assert d(0xF000, 0xFF00) == 0x0F00
    assert d(max_id, 0) == 1
    assert d(max_id, 1) == 2
    assert d(0, max_id) == -1
    assert d(1, max_id) == -2


def test_crc64_matches_stored(oracle):
    O = oracle
    cal, pf, pk = _load(O, "crc_test")
    assert pk.shape[0] == 34
    for p in pk:
        stored = int(np.frombuffer(p[-8:].tobytes(), dtype="<u8")[0])
        assert O.lib().ora_crc64(p.ctypes.data, p.size - 8) == stored


def test_geometry_matches_fixture_sizes(oracle):
    # SURVEY.md section 8 table; sizes cross-checked with the captures
    O = oracle
    for prof, h, size in [("RNG19_RFL8_SIG16_NIR16", 128, 24832), ("RNG15_RFL8_NIR8", 128, 8448),
                          ("RNG19_RFL8_SIG16_NIR16_DUAL", 32, 8448),
                          ("RNG15_RFL8_NIR8_DUAL", 128, 16640), ("LEGACY", 32, 6464),
                          ("LEGACY", 64, 12608), ("RNG19_RFL8_SIG16_NIR16_DUAL", 128, 33024)]:
        assert O.packet_format(prof, h, 16, 1024).lidar_packet_size == size


def test_legacy_col_status_is_all_ones(oracle):
    """tests/frame_batcher_test.cpp:749-769 (FrameBatcherLegacyTest.legacy_col_status): every column of
    the LEGACY capture reports status 0xFFFFFFFF (the last 4 bytes of a legacy column)."""
    O = oracle
    base = "OS-2-32-U0_v2.0.0_1024x10"
    cal = O.calib_from_json(os.path.join(PCAPS, base + ".json"))
    pf = cal.packet_format()
    pk = O.lidar_packets_from_pcap(os.path.join(PCAPS, base + ".pcap"), pf)
    assert len(pk) >= 64
    for p in pk:
        assert (O.packet_header(pf, "STATUS", p) == 0xFFFFFFFF).all()
