"""CPU-only property tests of the oracle, ported from the reference's own suites so the
checker is itself checked beyond the golden vectors:
  tests/packet_format_test.cpp:218-406    encode -> decode identity, dropped packets
  tests/parsing_benchmark_test.cpp:119-169  col path == block path
  tests/frame_batcher_test.cpp:73-303     invalid columns, custom planes untouched
  tests/destagger_test.cpp:135-210 / python/tests/test_destagger.py:86-114
  python/tests/test_xyzlut.py:119-136 + python/src/ouster/sdk/examples/reference.py:19-70
  tests/cartesian_test.cpp:53-99
"""
from math import cos, pi, sin, sqrt

import numpy as np
import pytest

ROUNDTRIP = [("LEGACY", 0), ("RNG19_RFL8_SIG16_NIR16", 0), ("RNG19_RFL8_SIG16_NIR16", 1),
             ("RNG19_RFL8_SIG16_NIR16_DUAL", 0), ("RNG15_RFL8_NIR8", 0), ("RNG15_RFL8_NIR8", 1),
             ("FUSA_RNG15_RFL8_NIR8_DUAL", 1), ("RNG15_RFL8_NIR8_DUAL", 0), ("FIVE_WORD_PIXEL", 0),
             ("RNG19_RFL8_SIG16_NIR16_RGB16_DUAL", 0), ("RNG19_RFL8_SIG16_ZONE16_DUAL", 0)]


@pytest.mark.parametrize("profile,hdr", ROUNDTRIP)
@pytest.mark.parametrize("force_col", [False, True])
def test_encode_decode_identity(oracle, profile, hdr, force_col):
    O = oracle
    cal = O.synthetic_calib(h=128, w=1024, profile=profile, header_type=hdr)
    pf = cal.packet_format()
    packets, src = O.synth_packets(cal, 1, with_window=True)
    assert packets.shape[1] == 64
    fr = O.Frame.for_profile(cal.profile, 128, 1024, 16, with_window=True)
    fr.fill(0x11)
    assert O.batch_frame(pf, packets[0], fr, init_id=cal.init_id & 0xFFFFFF, force_col=force_col)
    for n in fr.plane_names():
        assert np.array_equal(fr.plane(n), src[0].plane(n)), n
    assert np.array_equal(fr.timestamp, src[0].timestamp)
    assert np.array_equal(fr.status, src[0].status)
    assert fr.frame_id == src[0].frame_id


def test_dropped_and_invalid_columns(oracle):
    O = oracle
    cal = O.synthetic_calib(h=64, w=512, profile="RNG15_RFL8_NIR8_DUAL")
    pf = cal.packet_format()
    packets, src = O.synth_packets(cal, 1)
    pk = packets[0].copy()
    for c in (1, 5):  # invalidate two columns of packet 9
        pk[9, pf.packet_header_size + c * pf.col_size + 10] &= 0xFE
    keep = [i for i in range(32) if i not in (3, 20)]
    fr = O.Frame.for_profile(cal.profile, 64, 512, 16, with_window=True)
    fr.add_plane("CUSTOM0", O.U8)
    fr.fill(1)
    b = O.Batcher(pf, init_id=cal.init_id & 0xFFFFFF, expected_packets=len(keep))
    for i in keep:
        b.batch(pk[i], 100 + i, fr)
    rng = fr.plane("RANGE")
    want = src[0].plane("RANGE").copy()
    dead = list(range(48, 64)) + list(range(320, 336)) + [9 * 16 + 1, 9 * 16 + 5]
    want[:, dead] = 0
    assert np.array_equal(rng, want)
    assert np.all(fr.status[dead] == 0) and np.all(fr.timestamp[dead] == 0)
    assert np.all(fr.plane("CUSTOM0") == 1)            # planes unknown to the format untouched
    assert fr.packet_timestamp[3] == 0 and fr.packet_timestamp[9] == 109


def test_out_of_order_and_late_packets(oracle):
    O = oracle
    cal = O.synthetic_calib(h=32, w=512, profile="RNG15_RFL8_NIR8")
    pf = cal.packet_format()
    packets, src = O.synth_packets(cal, 3)
    fr = O.Frame.for_profile(cal.profile, 32, 512, 16)
    b = O.Batcher(pf, init_id=cal.init_id & 0xFFFFFF)
    order = list(range(32))
    order[4], order[5] = order[5], order[4]            # swapped inside a frame
    done = [b.batch(packets[0][i], 1 + i, fr) for i in order]
    assert done[-1] and sum(done) == 1
    assert np.array_equal(fr.plane("RANGE"), src[0].plane("RANGE"))
    assert not b.batch(packets[0][7], 99, fr) and b.dropped == 1   # late packet of a finished frame
    # next frame with one packet missing is released once 4 packets of the following frame queue up
    done = [b.batch(packets[1][i], 1 + i, fr) for i in range(32) if i != 6]
    assert not any(done)
    rel = [b.batch(packets[2][i], 1 + i, fr) for i in range(4)]
    assert rel == [False, False, False, True] and fr.frame_id == src[1].frame_id
    assert np.all(fr.plane("RANGE")[:, 96:112] == 0)


def test_destagger_is_roll_and_round_trips(oracle):
    O = oracle
    rng = np.random.default_rng(5)
    for dtype in (np.uint8, np.uint16, np.uint32, np.float64):
        img = rng.integers(0, 200, size=(64, 1024)).astype(dtype)
        shifts = rng.integers(-30, 31, 64)
        d = O.destagger(img, shifts)
        assert d.dtype == img.dtype and d.shape == img.shape
        want = np.stack([np.roll(img[u], shifts[u]) for u in range(64)])   # reference.py:131-158
        assert np.array_equal(d, want)
        assert np.array_equal(O.destagger(d, shifts, inverse=True), img)
    with pytest.raises(ValueError, match="image height does not match shifts size"):
        O.destagger(img, shifts[:-1])
    # non power-of-two widths follow the reference's unsigned arithmetic, not np.roll
    img = np.arange(4 * 1000, dtype=np.uint32).reshape(4, 1000)
    d = O.destagger(img, np.array([3, -3, 0, -1]))
    assert np.array_equal(d[0], np.roll(img[0], 3)) and np.array_equal(d[2], img[2])
    off = (1000 + ((1 << 64) - 3) % 1000) % 1000
    assert np.array_equal(d[1], np.roll(img[1], off))


def test_xyz_lut_matches_manual_formula(oracle):
    """reference.py:19-70 evaluated per pixel; rtol 1e-5 / atol 1e-8 like test_xyzlut.py:119-136."""
    O = oracle
    import os
    from conftest import PCAPS
    cal = O.calib_from_json(os.path.join(PCAPS, "OS-2-128-U1_v2.3.0_1024x10.json"))
    cal.beam_to_lidar[2, 3] = 1.25  # exercise the n = sqrt(bx^2 + bz^2) branch too
    ldir, lofs = cal.xyz_lut(False)
    rng = np.random.default_rng(1)
    r = rng.integers(1, 2 ** 19, size=(cal.h, cal.w)).astype(np.uint32)
    r[::7, ::5] = 0
    pts = O.cartesian(r, ldir, lofs).reshape(cal.h, cal.w, 3)
    bx, bz = cal.beam_to_lidar[0, 3], cal.beam_to_lidar[2, 3]
    n = sqrt(bx ** 2 + bz ** 2)
    worst = 0.0
    for u in range(0, cal.h, 9):
        for v in range(0, cal.w, 37):
            rr = float(r[u, v])
            if rr == 0:
                assert np.all(pts[u, v] == 0)
                continue
            te = 2.0 * pi * (1.0 - v / cal.w)
            ta = -2.0 * pi * (cal.beam_azimuth_angles[u] / 360.0)
            ph = 2.0 * pi * (cal.beam_altitude_angles[u] / 360.0)
            x = (rr - n) * cos(te + ta) * cos(ph) + bx * cos(te)
            y = (rr - n) * sin(te + ta) * cos(ph) + bx * sin(te)
            z = (rr - n) * sin(ph) + bz
            want = (cal.lidar_to_sensor @ np.array([x, y, z, 1.0]))[:3] * 0.001
            assert np.allclose(pts[u, v], want, rtol=1e-5, atol=1e-8)
            worst = max(worst, np.abs(pts[u, v] - want).max())
    assert worst < 1e-9


def test_float_lut_vs_double_and_extrinsics(oracle):
    O = oracle
    cal = O.synthetic_calib(h=64, w=512)
    ext = np.eye(4)
    ext[:3, 3] = [10.0, -2.0, 0.5]  # metres
    cal.extrinsic = ext
    d0, o0 = cal.xyz_lut(False)
    d1, o1 = cal.xyz_lut(True)
    assert np.allclose(d0, d1) and np.allclose(o1 - o0, [10.0, -2.0, 0.5])
    r = np.random.default_rng(2).integers(0, 2 ** 19, size=(64, 512)).astype(np.uint32)
    p64 = O.cartesian(r, d0, o0)
    p32 = O.cartesian(r, d0.astype(np.float32), o0.astype(np.float32))
    assert p32.dtype == np.float32
    assert np.abs(p32.astype(np.float64) - p64).max() < 1e-4   # cartesian_test.cpp: rel 1e-5
    with pytest.raises(ValueError, match="unexpected image dimensions"):
        O.cartesian(r[:32], d0, o0)
    with pytest.raises(ValueError, match="unexpected frame dimensions"):
        O.make_xyz_lut(512, 64, 0.001, np.eye(4), np.eye(4), np.zeros(63), np.zeros(63))
    with pytest.raises(ValueError, match="lut dimensions must be greater than zero"):
        O.make_xyz_lut(0, 64, 0.001, np.eye(4), np.eye(4), np.zeros(64), np.zeros(64))


def test_dewarp_frame_matches_dense_composition(oracle):
    """impl/dewarp_impl.h:23-81 restated == gate(dewarp(cartesian(range))) walked column-major over
    first_valid..last_valid (status & 1), skipping status == 0; empty for frames without valid columns."""
    O = oracle
    h, w = 16, 64
    cal = O.synthetic_calib(h=h, w=w, cpp=16)
    d, o = cal.xyz_lut(True)
    g = np.random.default_rng(3)
    r = g.integers(0, 60000, (h, w)).astype(np.uint32)
    r[g.random(r.shape) < 0.25] = 0
    st = np.ones(w, np.uint32)
    st[:3] = 0; st[10] = 0; st[11] = 2; st[60:] = 0
    ts = (np.arange(w) + 1000).astype(np.uint64)
    poses = np.tile(np.eye(4), (w, 1, 1))
    ang = g.uniform(-0.2, 0.2, w)
    poses[:, 0, 0] = np.cos(ang); poses[:, 0, 1] = -np.sin(ang)
    poses[:, 1, 0] = np.sin(ang); poses[:, 1, 1] = np.cos(ang)
    poses[:, :3, 3] = g.uniform(-4, 4, (w, 3))
    for ldt, tol in ((np.float64, 1e-12), (np.float32, 5e-5)):
        p, c, t = O.dewarp_frame(r, st, ts, poses, d.astype(ldt), o.astype(ldt), 1.0, 30.0)
        dense = O.dewarp(O.cartesian(r, d.astype(ldt), o.astype(ldt)), poses, h, w).reshape(h, w, 3)
        exp, ec = [], []
        for x in range(3, 60):
            if st[x] == 0:
                continue
            for y in range(h):
                if 1000 <= r[y, x] <= 30000:
                    exp.append(dense[y, x]); ec.append(x)
        assert np.array_equal(c, np.array(ec, np.uint32)) and np.array_equal(t, ts[c])
        assert p.dtype == ldt and np.abs(p.astype(np.float64) - np.array(exp, np.float64)).max() <= tol
    # column 11 (status 2) is inside the valid span and emitted; no valid column at all -> nothing
    assert 11 in c
    p, c, t = O.dewarp_frame(r, np.zeros(w, np.uint32), ts, poses, d, o, 0.0, 1000.0)
    assert len(p) == 0 and len(c) == 0
    # min > max and sub-millimetre windows keep nothing but zero-range... (ceil/floor of the bounds)
    p, _, _ = O.dewarp_frame(r, st, ts, poses, d, o, 0.0005, 0.0009)
    assert len(p) == 0
