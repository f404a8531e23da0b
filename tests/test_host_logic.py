"""CPU-only: the product's host-side format tables agree with the oracle's restatement of the
reference tables for every profile, and bench.py's synthetic workload is what it claims."""
import ctypes as C

import numpy as np
import pytest

from ouster_sdk_amd import _capi as capi

PROFILES = ["LEGACY", "RNG19_RFL8_SIG16_NIR16_DUAL", "RNG19_RFL8_SIG16_NIR16", "RNG15_RFL8_NIR8",
            "FIVE_WORD_PIXEL", "FUSA_RNG15_RFL8_NIR8_DUAL", "RNG15_RFL8_NIR8_DUAL",
            "RNG15_RFL8_NIR8_ZONE16", "RNG19_RFL8_SIG16_NIR16_ZONE16", "RNG15_RFL8_WIN8",
            "RNG19_RFL8_SIG16_ZONE16_DUAL", "RNG19_RFL8_SIG16_NIR16_RGB16",
            "RNG19_RFL8_SIG16_NIR16_RGB16_DUAL"]


@pytest.mark.parametrize("profile", PROFILES)
@pytest.mark.parametrize("header_type", [0, 1])
def test_format_desc_matches_oracle_tables(oracle, profile, header_type):
    O = oracle
    h, cpp, w = 64, 16, 1024
    pf = O.packet_format(profile, h, cpp, w, header_type)
    names = pf.field_names()
    fields = [(n, 8 if pf.field(n).num_elements == 1 else 6) for n in names]
    d = capi.format_desc(profile, h, cpp, w, fields, header_type)
    for a, b in [("packet_header_size", "packet_header_size"), ("col_header_size", "col_header_size"),
                 ("channel_data_size", "channel_data_size"), ("col_footer_size", "col_footer_size"),
                 ("packet_footer_size", "packet_footer_size"), ("col_size", "col_size"),
                 ("lidar_packet_size", "lidar_packet_size")]:
        assert getattr(d, a) == getattr(pf, b), a
    for i, n in enumerate(names):
        f = pf.field(n)
        got = d.fields[i].bits
        assert (got.offset, got.mask, got.shift) == (f.offset, f.mask, f.shift), n
    for dn, on in [("col_timestamp", "col_timestamp_info"), ("col_measurement_id", "col_measurement_id_info"),
                   ("col_status", "col_status_info"), ("frame_id", "frame_id_info"),
                   ("alert_flags", "alert_flags_info"), ("thermal_shutdown", "thermal_shutdown_info"),
                   ("shot_limiting", "shot_limiting_info"),
                   ("countdown_thermal_shutdown", "countdown_thermal_shutdown_info"),
                   ("countdown_shot_limiting", "countdown_shot_limiting_info")]:
        g, o = getattr(d, dn), getattr(pf, on)
        assert (g.offset, g.mask, g.shift) == (o.offset, o.mask, o.shift), dn


@pytest.mark.parametrize("profile", PROFILES)
def test_default_planes_match_oracle(oracle, profile):
    O = oracle
    fr = O.Frame.for_profile(profile, 16, 512, 16, with_window=True)
    want = {n: (fr.plane(n).itemsize * (fr.plane(n).shape[2] if fr.plane(n).ndim == 3 else 1))
            for n in fr.plane_names()}
    got = dict(capi.default_planes(profile, True))
    assert got == want


def test_bench_workload_definition(oracle):
    import bench
    O = oracle
    assert bench.algorithmic_bytes_per_frame() == 14_974_976
    pk = bench.synth_packets(1)
    assert pk.shape == (1, 128, 16640)
    cal = O.synthetic_calib(h=128, w=2048, profile="RNG15_RFL8_NIR8_DUAL")
    pf = cal.packet_format()
    fr = O.Frame.for_profile(cal.profile, 128, 2048, 16, with_window=True)
    assert O.batch_frame(pf, pk[0], fr, init_id=0x123456)
    assert fr.frame_id == 700 and np.all(fr.status == 1)
    assert np.array_equal(fr.measurement_id, np.arange(2048, dtype=np.uint16))
    z = (fr.plane("RANGE") == 0).mean()
    assert 0.25 < z < 0.35                                       # ~30 % no-return pixels
    assert fr.plane("RANGE").max() <= 0x3FFF8 and fr.plane("NEAR_IR").max() <= 0xFF0
    alt, az, shifts, b2l, l2s = bench.synth_calibration()
    assert np.array_equal(shifts, cal.pixel_shift_by_row)
    assert np.allclose(b2l, cal.beam_to_lidar) and np.allclose(l2s, cal.lidar_to_sensor)


def test_bench_gpus_flag_plans_the_launch():
    """`bench.py --gpus N`: one rank per GPU either way -- under a launcher WORLD_SIZE must equal N, without one the
    script starts N ranks itself through torch.distributed.run on 127.0.0.1."""
    import sys
    import bench
    assert bench.plan_launch(1, {}, [], 1) == ("run", 1)
    assert bench.plan_launch(8, {"WORLD_SIZE": "8", "RANK": "3"}, [], 8) == ("run", 8)
    with pytest.raises(SystemExit, match="must agree"):
        bench.plan_launch(8, {"WORLD_SIZE": "1"}, [], 8)
    with pytest.raises(SystemExit, match="must agree"):
        bench.plan_launch(1, {"WORLD_SIZE": "2"}, [], 8)
    with pytest.raises(SystemExit, match="only 1 GPU"):
        bench.plan_launch(2, {}, [], 1)
    with pytest.raises(SystemExit):
        bench.plan_launch(0, {}, [], 1)
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    action, cmd = bench.plan_launch(4, {}, argv, 8)
    assert action == "spawn" and cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-len(argv) - 1].endswith("bench.py") and cmd[-len(argv):] == argv
    # the one-device test knob lets a 1-GPU box exercise the N > 1 launch
    assert bench.plan_launch(2, {"BENCH_ONE_DEVICE": "1"}, argv, 1)[0] == "spawn"
