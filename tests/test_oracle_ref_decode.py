"""The oracle's packet decode pinned on the REFERENCE's own loop, compiled here from where it lies (oracle/Makefile, target
_ref/libdecode_ref.so): PacketFormat::block_field<T, BlockDim> (ouster_core/src/parsing.cpp:628-657) over
FieldDecodeInfo::get<T> (include/ouster/core/field_decode_info.h:41-54).  The geometry and field tables handed to it are the
oracle's (pinned on the reference's bit-width table and header KATs in test_oracle_golden.py); what is checked here is the
decode loop itself: ora_block_field == block_field for every field of every static profile, on random packets, for every
block size the reference uses, at measurement ids that are not packet-aligned, and where the reference throws."""
import numpy as np
import pytest

from oracle import decode_ref

pytestmark = pytest.mark.skipif(not decode_ref.available(), reason="oracle/_ref/libdecode_ref.so is built where /root/reference exists")

PROFILES = ["LEGACY", "RNG19_RFL8_SIG16_NIR16_DUAL", "RNG19_RFL8_SIG16_NIR16", "RNG15_RFL8_NIR8", "RNG15_RFL8_NIR8_DUAL",
            "FIVE_WORD_PIXEL", "FUSA_RNG15_RFL8_NIR8_DUAL"]


def _decode_both(O, pf, rpf, name, packet, h, w, elem_dtype, bd):
    a = np.full((h, w), 0xAB, dtype=elem_dtype)
    b = np.full((h, w), 0xAB, dtype=elem_dtype)
    rc_ref = rpf.block_field(a, name, packet, bd)
    import ctypes as C
    rc_ora = O.lib().ora_block_field(C.byref(pf), b.ctypes.data, b.itemsize, w, name.encode(), packet.ctypes.data, bd)
    return rc_ref, rc_ora, a, b


@pytest.mark.parametrize("profile", PROFILES)
def test_oracle_block_field_equals_the_references(oracle, profile):
    O = oracle
    try:
        cal = O.synthetic_calib(h=32, w=256, profile=profile)
    except Exception:
        pytest.skip("profile not in the oracle's table")
    pf = cal.packet_format()
    rpf = decode_ref.RefPacketFormat(O, pf)
    packets, src = O.synth_packets(cal, 2, seed=0xBEEF)
    rng = np.random.default_rng(1)
    for name in rpf.names:
        if name not in src[0].plane_names():
            continue
        plane = src[0].plane(name)
        if plane.ndim != 2:
            continue
        for bd in (16, 8, 4):
            if pf.columns_per_packet % bd or pf.pixels_per_column % bd:
                continue
            for p in rng.integers(0, packets.shape[1], 3):
                rc_ref, rc_ora, a, b = _decode_both(O, pf, rpf, name, packets[1, int(p)], cal.h, cal.w, plane.dtype, bd)
                assert rc_ref == 0 and rc_ora == 0, (name, bd, rc_ref, rc_ora)
                assert np.array_equal(a, b), (name, bd, int(p))
                c0 = int(p) * cal.cpp
                assert np.array_equal(a[:, c0:c0 + cal.cpp], src[1].plane(name)[:, c0:c0 + cal.cpp]), (name, bd)


def test_wide_destination_and_too_small_destination(oracle):
    """T wider than the field zero-extends (memcpy of the masked 64-bit word); T narrower than the field's type throws
    std::invalid_argument("Dest type too small for specified field") -- both like the reference."""
    O = oracle
    cal = O.synthetic_calib(h=16, w=64, profile="RNG19_RFL8_SIG16_NIR16")
    pf = cal.packet_format()
    rpf = decode_ref.RefPacketFormat(O, pf)
    packets, src = O.synth_packets(cal, 1)
    for name, wide, small in (("REFLECTIVITY", np.uint32, None), ("RANGE", np.uint64, np.uint16), ("SIGNAL", np.uint32, np.uint8)):
        rc_ref, rc_ora, a, b = _decode_both(O, pf, rpf, name, packets[0, 1], cal.h, cal.w, wide, 16)
        assert rc_ref == 0 and rc_ora == 0 and np.array_equal(a, b)
        assert np.array_equal(a[:, 16:32], src[0].plane(name)[:, 16:32].astype(wide))
        if small is not None:
            rc_ref, rc_ora, _, _ = _decode_both(O, pf, rpf, name, packets[0, 1], cal.h, cal.w, small, 16)
            assert rc_ref == -2 and rc_ora == -2


def test_unaligned_measurement_ids(oracle):
    """block_field writes a block at the measurement id of its FIRST column, whatever that is (parsing.cpp:647-652): rewrite
    the ids of a packet and compare."""
    O = oracle
    cal = O.synthetic_calib(h=16, w=128, profile="RNG15_RFL8_NIR8_DUAL")
    pf = cal.packet_format()
    rpf = decode_ref.RefPacketFormat(O, pf)
    packets, _ = O.synth_packets(cal, 1)
    pk = packets[0, 2].copy()
    for ic in range(cal.cpp):   # ids 37, 38, ... (not a multiple of 16; the block path only reads the first id of a block)
        off = pf.packet_header_size + ic * pf.col_size + 8
        pk[off:off + 2] = np.frombuffer(np.uint16(37 + ic).tobytes(), np.uint8)
    for bd in (16, 8, 4):
        rc_ref, rc_ora, a, b = _decode_both(O, pf, rpf, "RANGE", pk, cal.h, cal.w, np.uint32, bd)
        assert rc_ref == 0 and rc_ora == 0 and np.array_equal(a, b)
        assert (a[:, 37:53] != 0xAB).any() and (a[:, :37] == 0xAB).all() and (a[:, 53:] == 0xAB).all()
