"""GPU: the C++ FrameBatcher mirror (via ouster_sdk_amd.core) against the oracle's restatement
of the reference state machine on randomized packet streams -- drops, swaps inside and across
frame boundaries, duplicates, late packets, frame-id wrap-around, sensor re-initialisation
(init_id change).  Both must release the same frames at the same packets with identical
content.  Reference behaviour: ouster_core/src/lidar_frame.cpp:1743-1927; the scenarios are
those of tests/frame_batcher_test.cpp:73-545."""
import numpy as np
import pytest

from conftest import has_gpu

pytestmark = pytest.mark.gpu

if has_gpu():
    from ouster_sdk_amd import core

H, W, CPP = 16, 256, 16
PPF = W // CPP


def _info(profile, header_type, init_id):
    info = core.SensorInfo()
    f = info.format
    f.pixels_per_column, f.columns_per_frame, f.columns_per_packet = H, W, CPP
    f.column_window = (0, W - 1)
    f.udp_profile_lidar = core.UDPProfileLidar.from_string(profile)
    f.header_type = core.HeaderType.FUSA if header_type else core.HeaderType.STANDARD
    f.pixel_shift_by_row = [0] * H
    info.format = f
    info.beam_azimuth_angles = [0.0] * H
    info.beam_altitude_angles = [0.0] * H
    info.beam_to_lidar_transform = np.eye(4)
    info.lidar_to_sensor_transform = np.eye(4)
    info.init_id = init_id
    info.fw_rev = "v3.2.0"
    return info


def _stream(O, rng, profile, header_type, scenario):
    """List of (packet bytes, host_ts) plus the init id the batchers start with."""
    cal = O.synthetic_calib(h=H, w=W, cpp=CPP, profile=profile, header_type=header_type)
    pf = cal.packet_format()
    init0 = 0x0ABCDE
    max_id = 0xFFFFFFFF if header_type else 0xFFFF
    first = max_id - 2 if scenario == "wrap" else 100
    frames = []
    for k in range(6):
        fr = O.Frame.for_profile(cal.profile, H, W, CPP, with_window=True)
        O.randomize_frame(fr, pf, 1000 + k, frame_id=(first + k) & max_id)
        init = init0 if not (scenario == "reinit" and k >= 3) else init0 + 1
        pk, _ = O.frame_to_packets(fr, pf, init, 7)
        frames.append([p.copy() for p in pk])
    seq = []
    for k, pk in enumerate(frames):
        order = list(range(PPF))
        if scenario == "drops":
            for d in rng.choice(PPF, size=rng.integers(1, 4), replace=False):
                order.remove(int(d))
        if scenario == "swaps":
            i = int(rng.integers(0, PPF - 1))
            order[i], order[i + 1] = order[i + 1], order[i]
        if scenario == "dups":
            order.insert(int(rng.integers(1, PPF)), int(rng.integers(0, PPF)))
        seq += [(k, i) for i in order]
    if scenario == "boundary":       # last packets of frame k arrive after the first of k+1
        for k in range(5):
            idx = [j for j, (fk, _) in enumerate(seq) if fk == k][-1]
            nxt = [j for j, (fk, _) in enumerate(seq) if fk == k + 1][:int(rng.integers(1, 4))]
            item = seq.pop(idx)
            seq.insert(nxt[-1], item)
    if scenario == "late":           # a packet of frame 1 shows up in the middle of frame 3
        idx = [j for j, (fk, _) in enumerate(seq) if fk == 1][3]
        item = seq.pop(idx)
        pos = [j for j, (fk, _) in enumerate(seq) if fk == 3][5]
        seq.insert(pos, item)
    return cal, pf, init0, [(frames[k][i], 1 + n) for n, (k, i) in enumerate(seq)]


@pytest.mark.parametrize("profile,header_type", [("RNG15_RFL8_NIR8_DUAL", 0), ("FUSA_RNG15_RFL8_NIR8_DUAL", 1),
                                                 ("LEGACY", 0)])
@pytest.mark.parametrize("scenario", ["clean", "drops", "swaps", "dups", "boundary", "late", "wrap", "reinit"])
def test_mirror_batcher_equals_oracle_state_machine(oracle, profile, header_type, scenario):
    O = oracle
    if profile == "LEGACY" and scenario == "reinit":
        pytest.skip("legacy packets carry no init id")
    rng = np.random.default_rng(hash((profile, scenario)) & 0xFFFF)
    cal, pf, init0, stream = _stream(O, rng, profile, header_type, scenario)
    info = _info(profile, header_type, init0)
    cpf = core.PacketFormat(info)
    mirror, mframe = core.FrameBatcher(info), core.LidarFrame(info)
    orc, oframe = O.Batcher(pf, init_id=init0), O.Frame.for_profile(cal.profile, H, W, CPP, with_window=True)
    released = 0
    for n, (buf, ts) in enumerate(stream):
        lp = core.LidarPacket(cpf.lidar_packet_size)
        lp.buf = buf.tobytes()
        lp.host_timestamp = ts
        try:
            want = orc.batch(buf, ts, oframe)
            oerr = None
        except RuntimeError as e:
            want, oerr = None, e
        try:
            got = mirror(lp, mframe)
            merr = None
        except RuntimeError as e:
            got, merr = None, e
        assert (oerr is None) == (merr is None), (n, oerr, merr)
        if oerr is not None:
            break
        assert got == want, (scenario, n)
        if want:
            released += 1
            assert mframe.frame_id == oframe.frame_id
            for name in oframe.plane_names():
                assert np.array_equal(mframe.field(name), oframe.plane(name)), (scenario, n, name)
            assert np.array_equal(mframe.timestamp, oframe.timestamp)
            assert np.array_equal(mframe.status, oframe.status)
            assert np.array_equal(mframe.measurement_id, oframe.measurement_id)
            assert np.array_equal(mframe.packet_timestamp, oframe.packet_timestamp)
            assert np.array_equal(mframe.alert_flags, oframe.alert_flags)
    assert released >= 3
    assert mirror.dropped_packets() == orc.dropped
