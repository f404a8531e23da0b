"""GPU parity for the optimistic decode pass + fix-up pass and for the sizes that
bench.py actually runs (the benchmarked batch was never checked).

  * home-slot frames with holes / invalid columns / out-of-range ids never leave the fast pass
    (asserted by switching the fix-up pass OFF) and equal the oracle;
  * compacted drops, shuffles and duplicates are caught on the device and redone by the fix-up pass;
  * the call keeps no state: alternating workloads, a captured graph replayed on changing loss
    patterns with eager calls in between (stale column map on replay);
  * packet_counts as a device tensor, a pinned host tensor and a plain numpy array;
  * configs[2] at 256 frames, configs[1] at 512 frames, configs[4]'s 4-sensor batch at W = 2048 through
    the narrow and the 256-wide kernels: frames {0, 7, 8, 255, ...} against the oracle.
Mirrors tests/frame_batcher_test.cpp:73-303 (dropped / invalid / custom) and
tests/packet_format_test.cpp:218-326 (encode -> decode identity) of the reference.
"""
import numpy as np
import pytest

from conftest import has_gpu

pytestmark = pytest.mark.gpu

if has_gpu():
    import torch
    from ouster_sdk_amd.device import HotPath

from test_gpu_parity import _np, _oracle_frames  # noqa: E402

# "s128" / "s256": the persistent, double-buffered k_decode_stream with that tile width
VARIANTS = [("narrow", 0), ("wide64", 64), ("wide128", 128), ("wide256", 256), ("stream128", "s128"),
            ("stream256", "s256")]


def _force_variant(hp, wide):
    """wide: None = the library's choice; 0 = k_decode; 64..512 = k_decode_wide; "s128" / "s256" = k_decode_stream."""
    if wide is None:
        return
    if isinstance(wide, str):
        hp.ctx.set_knob("stream", int(wide[1:]))
        hp.ctx.set_knob("stream_min_tiles", 0)
        return
    hp.ctx.set_knob("stream", 0)
    hp.ctx.set_knob("wide", wide)
    hp.ctx.set_knob("wide_min_blocks", 0)


def _assert_variant_ran(hp, wide):
    tc, _ = hp.ctx.last_decode_tile()
    kernel = hp.ctx.last_decode_kernel()
    if isinstance(wide, str):
        assert kernel in ("k_decode_stream", "k_decode_stream2") and tc == int(wide[1:]), (kernel, tc, wide)
    elif wide:
        assert kernel == "k_decode_wide" and tc == wide, (kernel, tc, wide)
    elif wide == 0:
        assert kernel == "k_decode" and tc <= 64, (kernel, tc, wide)


def _hotpath(cal, profile, wide=None, fixup=True, use_extrinsics=False, **kw):
    hp = HotPath(profile, cal.h, cal.w, cal.cpp, header_type=cal.header_type, **kw)
    hp.set_pixel_shift_by_row(cal.pixel_shift_by_row)
    hp.add_lut(cal.beam_to_lidar, cal.lut_transform(use_extrinsics), cal.beam_azimuth_angles,
               cal.beam_altitude_angles)
    _force_variant(hp, wide)
    if not fixup:
        hp.ctx.set_knob("fixup", 0)
    return hp


def _compare(O, cal, hp, out, ref_frames, dst_names, xyz_names, use_extrinsics=False, frames=None,
             check_nvalid=True):
    ldir, lofs = cal.xyz_lut(use_extrinsics)
    names = [n for n, _ in hp.fields]
    worst = 0.0
    for f, fr in (enumerate(ref_frames) if frames is None else frames):
        for n in names:
            if n in out:
                assert np.array_equal(_np(out[n][f]), fr.plane(n)), (f, n)
        assert np.array_equal(_np(out["timestamp"][f]), fr.timestamp), f
        assert np.array_equal(_np(out["measurement_id"][f]), fr.measurement_id), f
        assert np.array_equal(_np(out["status"][f]), fr.status), f
        for n in dst_names:
            assert np.array_equal(_np(out["destaggered:" + n][f]),
                                  O.destagger(fr.plane(n), cal.pixel_shift_by_row)), (f, n)
        for n in xyz_names:
            want = O.cartesian(fr.plane(n), ldir, lofs)
            err = np.abs(_np(out["xyz:" + n][f]).astype(np.float64) - want).max()
            worst = max(worst, float(err))
            assert err <= 1e-4, (f, n, err)
        meta = _np(out["frame_meta"][f])
        if (fr.status & 1).any():   # frame-level values come from the first packet RECEIVED, even when slot 0 is a hole
            assert meta[:8].view(np.int64)[0] == fr.frame_id, ("frame_id", f, meta[:8].view(np.int64)[0], fr.frame_id)
        if not check_nvalid:     # published by the fix-up pass
            continue
        assert meta[20:24].view(np.uint32)[0] == int((fr.status & 1).sum()), ("n_valid_columns", f)
    return worst


@pytest.mark.parametrize("label,wide", VARIANTS)
def test_home_slots_with_holes_stay_on_the_fast_pass(oracle, label, wide):
    """Packets in their home slots; lost packets are holes (zeroed slots), some columns invalid, one
    column with an out-of-range measurement id, one frame entirely empty.  With the fix-up pass
    switched off the result must still be the oracle's: the optimistic pass alone handles loss."""
    O = oracle
    cal = O.synthetic_calib(h=64, w=1024, profile="RNG15_RFL8_NIR8_DUAL")
    pf = cal.packet_format()
    n = 9
    packets, _ = O.synth_packets(cal, n, with_window=True)
    host = packets.copy()
    rng = np.random.default_rng(3)
    lost = {2: [5, 6, 40], 6: [0], 8: list(range(64))}       # frame 8: nothing arrived
    for f, ps in lost.items():
        host[f, ps] = 0
    for p in rng.integers(0, 64, 4):                          # invalid columns (status bit 0 clear)
        for c in rng.integers(0, 16, 5):
            host[4, p, pf.packet_header_size + c * pf.col_size + 10] &= 0xFE
    off = pf.packet_header_size + 3 * pf.col_size + 8          # a column beyond the frame: dropped
    host[5, 7, off:off + 2] = np.frombuffer(np.uint16(5000).tobytes(), np.uint8)
    by_frame = [np.delete(host[f], lost.get(f, []), axis=0) for f in range(n)]
    hp = _hotpath(cal, "RNG15_RFL8_NIR8_DUAL", wide=wide, fixup=False)
    dst, xyz = ["RANGE", "REFLECTIVITY2"], ["RANGE", "RANGE2"]
    out = hp.alloc_outputs(n, destagger=dst, xyz=xyz)
    for t in out.values():
        t.view(torch.uint8).fill_(0xCD)
    hp.decode(torch.from_numpy(host).cuda(), out)
    hp.sync()
    _assert_variant_ran(hp, wide)
    ref = _oracle_frames(O, cal, pf, by_frame, True)
    _compare(O, cal, hp, out, ref, dst, xyz, check_nvalid=False)
    hp.ctx.set_knob("fixup", 1)                               # the normal two-pass call: same bytes + counts
    for t in out.values():
        t.view(torch.uint8).fill_(0xCD)
    hp.decode(torch.from_numpy(host).cuda(), out)
    hp.sync()
    _compare(O, cal, hp, out, ref, dst, xyz)


@pytest.mark.parametrize("label,wide", VARIANTS)
def test_strays_are_caught_and_redone(oracle, label, wide):
    """One slot per column of the frame, but the slots are compacted after a drop / shuffled /
    carry a duplicate: the fast pass flags those frames, the fix-up pass redoes them; clean frames of
    the same batch stay untouched.  Without the fix-up pass the flagged frames are visibly undone."""
    O = oracle
    cal = O.synthetic_calib(h=32, w=1024, profile="RNG19_RFL8_SIG16_NIR16")
    pf = cal.packet_format()
    n = 10
    packets, _ = O.synth_packets(cal, n, with_window=True)
    rng = np.random.default_rng(11)
    by_frame = [packets[f] for f in range(n)]
    by_frame[1] = np.delete(by_frame[1], [9], axis=0)                     # compacted after a drop
    by_frame[3] = by_frame[3][rng.permutation(64)]                        # any order
    by_frame[5] = np.concatenate([np.delete(by_frame[5], [20], axis=0), by_frame[5][30:31]])  # duplicate
    sw = by_frame[7].copy(); sw[[10, 11]] = sw[[11, 10]]; by_frame[7] = sw   # two neighbours swapped
    host = np.zeros((n, 64, pf.lidar_packet_size), np.uint8)
    counts = np.zeros(n, np.uint32)
    for f, pk in enumerate(by_frame):
        host[f, :len(pk)] = pk
        counts[f] = len(pk)
    dev = torch.from_numpy(host).cuda()
    dst, xyz = ["RANGE"], ["RANGE"]
    ref = _oracle_frames(O, cal, pf, by_frame, True)
    for counts_arg in (counts, torch.from_numpy(counts.astype(np.int32)).cuda(),
                       torch.from_numpy(counts.astype(np.int32)).pin_memory()):
        hp = _hotpath(cal, "RNG19_RFL8_SIG16_NIR16", wide=wide)
        out = hp.alloc_outputs(n, destagger=dst, xyz=xyz)
        for t in out.values():
            t.view(torch.uint8).fill_(0xCD)
        hp.decode(dev, out, packet_counts=counts_arg)
        hp.sync()
        _compare(O, cal, hp, out, ref, dst, xyz)
    # detection really is what routes them: with the second pass off, exactly the odd frames are wrong
    hp = _hotpath(cal, "RNG19_RFL8_SIG16_NIR16", wide=wide, fixup=False)
    out = hp.alloc_outputs(n)
    for t in out.values():
        t.view(torch.uint8).fill_(0xCD)
    hp.decode(dev, out, packet_counts=counts)
    hp.sync()
    for f in range(n):
        same = np.array_equal(_np(out["RANGE"][f]), ref[f].plane("RANGE"))
        assert same == (f not in (1, 3, 5, 7)), f


def test_decode_keeps_no_state_between_calls(oracle):
    """Alternating full / compacted-partial / hole frames through ONE context for many calls (the r01
    build kept an epoch-tagged column map between calls)."""
    O = oracle
    cal = O.synthetic_calib(h=16, w=256, profile="RNG15_RFL8_NIR8")
    full, src = O.synth_packets(cal, 1, with_window=True)
    hp = HotPath("RNG15_RFL8_NIR8", 16, 256, 16)
    d_full = torch.from_numpy(full).cuda()
    d_part = torch.zeros_like(d_full)
    d_part[:, :11] = d_full[:, [0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 11]]   # packet 5 lost, compacted; 12..15 missing
    d_hole = d_full.clone()
    d_hole[:, 5] = 0
    cnt_part = torch.tensor([11], dtype=torch.int32, device="cuda")
    want_full = src[0].plane("RANGE")
    want_part = want_full.copy(); want_part[:, 80:96] = 0; want_part[:, 192:] = 0
    want_hole = want_full.copy(); want_hole[:, 80:96] = 0
    out = hp.alloc_outputs(1)
    for call in range(300):
        kind = call % 3
        if kind == 0:
            hp.decode(d_full, out)
        elif kind == 1:
            hp.decode(d_part, out, packet_counts=cnt_part)
        else:
            hp.decode(d_hole, out)
        if call < 9 or call % 37 == 0 or call >= 290:
            got = _np(out["RANGE"][0])
            assert np.array_equal(got, (want_full, want_part, want_hole)[kind]), call
            nv = _np(out["frame_meta"][0])[20:24].view(np.uint32)[0]
            assert nv == (256, 176, 240)[kind], (call, nv)


def test_graph_replay_on_changing_loss_patterns(oracle):
    """Capture ouster_hip_decode once, replay it on new packet contents in the same buffers: full
    frames, swapped packets (strays -> fix-up), holes, with an eager decode of the same context in
    between (the epoch map of the r01 build went stale in exactly these cases)."""
    O = oracle
    cal = O.synthetic_calib(h=64, w=512, profile="RNG15_RFL8_NIR8_DUAL")
    pf = cal.packet_format()
    variants = []
    for seed in (5, 6, 7, 8):
        pk, _ = O.synth_packets(cal, 2, seed=seed, with_window=True)
        variants.append(pk)
    swapped = variants[1].copy(); swapped[0, [3, 20]] = swapped[0, [20, 3]]
    holes = variants[2].copy(); holes[1, [0, 9, 31]] = 0
    contents = [("full", variants[0], [variants[0][0], variants[0][1]]),
                ("swapped", swapped, [swapped[0], swapped[1]]),
                ("holes", holes, [holes[0], np.delete(holes[1], [0, 9, 31], axis=0)]),
                ("full2", variants[3], [variants[3][0], variants[3][1]])]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        hp = _hotpath(cal, "RNG15_RFL8_NIR8_DUAL", use_extrinsics=True)
        d_pk = torch.from_numpy(variants[0]).cuda()
        d_other = torch.from_numpy(variants[3]).cuda()
        dst, xyz = ["RANGE"], ["RANGE", "RANGE2"]
        out = hp.alloc_outputs(2, destagger=dst, xyz=xyz)
        out_other = hp.alloc_outputs(2)
        hp.decode(d_pk, out)          # warm: scratch allocation, offsets / LUT descriptor upload
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            hp.decode(d_pk, out)
        for i, (label, pk, by_frame) in enumerate(contents):
            d_pk.copy_(torch.from_numpy(pk))
            for t in out.values():
                t.view(torch.uint8).fill_(0xEE)
            g.replay()
            if i % 2 == 0:
                g.replay()
            s.synchronize()
            ref = _oracle_frames(O, cal, pf, by_frame, True)
            _compare(O, cal, hp, out, ref, dst, xyz, use_extrinsics=True)
            hp.decode(d_other, out_other)   # an eager call between replays must not disturb them
            s.synchronize()
            assert np.array_equal(_np(out_other["status"][0]) & 1, np.ones(512, np.uint32)), label
    torch.cuda.synchronize()


# ---------------------------------------------------------------------------------------------
# the benchmarked sizes
# ---------------------------------------------------------------------------------------------
def _bench_size_case(O, profile, n, dst, xyz, wide, n_luts=1, check=(0, 7, 8, 255, 256, 511)):
    cal = O.synthetic_calib(h=128, w=2048, profile=profile)
    packets, src = O.synth_packets(cal, 8, with_window=True)     # 8 distinct frames
    hp = HotPath(profile, 128, 2048, 16)
    hp.set_pixel_shift_by_row(cal.pixel_shift_by_row)
    cals = []
    for k in range(n_luts):
        c = O.synthetic_calib(h=128, w=2048, profile=profile)
        if n_luts > 1:
            a = 0.5 * np.pi * k
            ext = np.eye(4)
            ext[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
            ext[:3, 3] = [1.0 * k, -0.5 * k, 0.25]
            c.extrinsic = ext
        cals.append(c)
        hp.add_lut(c.beam_to_lidar, c.lut_transform(n_luts > 1), c.beam_azimuth_angles, c.beam_altitude_angles)
    if isinstance(wide, str):
        _force_variant(hp, wide)
    elif wide is not None:
        hp.ctx.set_knob("stream", 0)
        hp.ctx.set_knob("wide", wide)
    d_pk = torch.from_numpy(packets).cuda().repeat(n // 8, 1, 1).contiguous()
    out = hp.alloc_outputs(n, destagger=dst, xyz=xyz)
    for t in out.values():
        t.view(torch.uint8).fill_(0xA5)
    hp.decode(d_pk, out)
    hp.sync()
    tc, tr = hp.ctx.last_decode_tile()
    if isinstance(wide, str):
        _assert_variant_ran(hp, wide)
    names = [x for x, _ in hp.fields]
    luts = [c.xyz_lut(n_luts > 1) for c in cals]
    worst = 0.0
    for f in [f for f in check if f < n]:
        fr = src[f % 8]
        for name in names:
            assert np.array_equal(_np(out[name][f]), fr.plane(name)), (f, name)
        assert np.array_equal(_np(out["timestamp"][f]), fr.timestamp), f
        assert np.array_equal(_np(out["status"][f]), fr.status), f
        for name in dst:
            assert np.array_equal(_np(out["destaggered:" + name][f]),
                                  O.destagger(fr.plane(name), cal.pixel_shift_by_row)), (f, name)
        d, o = luts[f % n_luts]
        for name in xyz:
            want = O.cartesian(fr.plane(name), d, o)
            err = np.abs(_np(out["xyz:" + name][f]).astype(np.float64) - want).max()
            worst = max(worst, float(err))
            assert err <= 1e-4, (f, name, err)
    # every other frame: identical inputs + identical LUT -> identical bytes (period lcm(8, n_luts) = 8)
    for k, v in out.items():
        if k == "frame_meta":
            continue
        v8 = v.view(torch.uint8).reshape(n // 8, 8, -1)
        assert torch.equal(v8, v8[:1].expand_as(v8)), k
    return tc, tr, worst


def test_bench_size_dual_256_frames(oracle):
    """configs[2] exactly as bench.py runs it: 256 frames of 128x2048 dual return, all outputs; the
    variant the tuner would pick from (256-wide) and the 64-column kernel."""
    dst = ["RANGE", "RANGE2", "REFLECTIVITY", "REFLECTIVITY2"]
    for wide, want_tc in ((256, 256), (0, 64), (128, 128), ("s256", 256), ("s128", 128)):
        tc, tr, worst = _bench_size_case(oracle, "RNG15_RFL8_NIR8_DUAL", 256, dst, ["RANGE", "RANGE2"], wide)
        assert tc == want_tc, (tc, tr)
        assert worst <= 4e-5


def test_bench_size_single_512_frames(oracle):
    """configs[1] / configs[3]: 512 frames of 128x2048 RNG19_RFL8_SIG16_NIR16 (the XCD-aware block ->
    frame map with 64 frames per XCD), default variant selection and the forced ones."""
    for wide in (None, 0, 128, "s128", "s256"):
        tc, tr, worst = _bench_size_case(oracle, "RNG19_RFL8_SIG16_NIR16", 512, ["RANGE", "REFLECTIVITY"],
                                         ["RANGE"], wide)
        assert worst <= 4e-5


def test_bench_size_fused4_w2048(oracle):
    """configs[4]'s per-GPU share as `bench.py --workload fused4` runs it: 4 sensors interleaved
    (frame % 4), each with its own extrinsics folded into its LUT, W = 2048, through the 256-wide
    kernel AND the narrow one."""
    dst = ["RANGE", "RANGE2", "REFLECTIVITY", "REFLECTIVITY2"]
    for wide, want_tc in ((256, 256), (0, 64), ("s256", 256)):
        tc, tr, worst = _bench_size_case(oracle, "RNG15_RFL8_NIR8_DUAL", 256, dst, ["RANGE", "RANGE2"], wide,
                                         n_luts=4)
        assert tc == want_tc
        assert worst <= 4e-5


@pytest.mark.parametrize("label,wide", VARIANTS)
def test_range_gate_counts_feed_the_dewarp(oracle, label, wide):
    """The decode kernels count, per column, the RANGE pixels inside a dewarp's range gate (by-product
    `gate_counts`); a dewarp_frames with those counts skips its counting pass and must give exactly the
    bytes of the stand-alone three-kernel dewarp -- on clean frames, frames with holes and frames that
    went through the fix-up pass."""
    O = oracle
    cal = O.synthetic_calib(h=64, w=1024, profile="RNG15_RFL8_NIR8_DUAL")
    pf = cal.packet_format()
    n = 6
    packets, src = O.synth_packets(cal, n, with_window=True)
    host = packets.copy()
    host[1, [3, 4]] = 0                                        # holes
    sw = host[2].copy(); sw[[10, 11]] = sw[[11, 10]]; host[2] = sw   # strays -> fix-up pass
    hp = _hotpath(cal, "RNG15_RFL8_NIR8_DUAL", wide=wide, use_extrinsics=True)
    out = hp.alloc_outputs(n, xyz=["RANGE"])
    gate = (2.0, 90.0)
    hp.decode(torch.from_numpy(host).cuda(), out, gate=gate)
    hp.sync()
    lo, hi, empty = hp.range_gate(*gate)
    assert (lo, hi, empty) == (2000, 90000, False)
    rng = out["RANGE"]
    want = ((rng.to(torch.int64) >= lo) & (rng.to(torch.int64) <= hi)).sum(dim=1)          # [n, W]
    got = out["gate_counts"].to(torch.int64).sum(dim=1)
    assert torch.equal(got, want)
    poses = torch.eye(4, dtype=torch.float64, device="cuda").repeat(n, cal.w, 1, 1).contiguous()
    poses[:, :, 0, 3] = torch.arange(cal.w, device="cuda", dtype=torch.float64) * 0.01
    a = hp.dewarp_frames(rng, out["status"], poses, *gate, timestamp=out["timestamp"])
    b = hp.dewarp_frames(rng, out["status"], poses, *gate, timestamp=out["timestamp"], gate_counts=out["gate_counts"])
    assert torch.equal(a["frame_offsets"], b["frame_offsets"])
    tot = int(a["frame_offsets"][-1].item())
    assert tot > 1000
    for k in ("points", "frame_idxs", "col_idxs", "timestamps_ns"):
        assert torch.equal(a[k][:tot].view(torch.uint8), b[k][:tot].view(torch.uint8)), k


def test_pick_placement_returns_equivalent_buffers(oracle):
    """HotPath.pick_placement draws candidate allocations, times the decode into each and keeps the fastest:
    the winners are ordinary buffers -- same names, dtypes and shapes as alloc_outputs, same decoded bytes."""
    O = oracle
    cal = O.synthetic_calib(h=64, w=1024, profile="RNG15_RFL8_NIR8_DUAL")
    packets, src = O.synth_packets(cal, 4, with_window=True)
    hp = _hotpath(cal, "RNG15_RFL8_NIR8_DUAL")
    d_pk = torch.from_numpy(packets).cuda().repeat(8, 1, 1).contiguous()
    make = lambda: hp.alloc_outputs(32, destagger=["RANGE", "REFLECTIVITY2"], xyz=["RANGE", "RANGE2"])  # noqa: E731
    plain = make()
    hp.decode(d_pk, plain)
    pk2, out, rep = hp.pick_placement(d_pk, make, tries=3, launches=2)
    assert rep["tries"] == 3 and len(rep["output_sets_ms"]) == 3 and len(rep["packet_buffers_ms"]) == 3
    assert torch.equal(pk2, d_pk)
    assert list(out) == list(plain)
    for k in plain:
        assert out[k].dtype == plain[k].dtype and out[k].shape == plain[k].shape and out[k].is_contiguous(), k
        assert out[k].data_ptr() % 256 == 0, k
    for t in out.values():
        t.view(torch.uint8).fill_(0x3C)
    hp.decode(pk2, out)
    hp.sync()
    for k in plain:
        assert torch.equal(out[k].view(torch.uint8), plain[k].view(torch.uint8)), k
    _compare(O, cal, hp, out, [src[f % 4] for f in range(32)], ["RANGE", "REFLECTIVITY2"], ["RANGE", "RANGE2"])


def test_placement_setup_sequence_keeps_the_bytes(oracle):
    """bench.py's round-6 setup: refine_placement, pick_placement with per-array draws and the caller's set as a candidate
    (slab=False, incumbent=out), refine_placement again.  Whatever wins is an ordinary set of buffers and decodes to the same
    bytes; the report names the caller's set's time."""
    O = oracle
    cal = O.synthetic_calib(h=64, w=1024, profile="RNG15_RFL8_NIR8_DUAL")
    packets, src = O.synth_packets(cal, 4, with_window=True)
    hp = _hotpath(cal, "RNG15_RFL8_NIR8_DUAL")
    d_pk = torch.from_numpy(packets).cuda().repeat(8, 1, 1).contiguous()
    make = lambda: hp.alloc_outputs(32, destagger=["RANGE", "REFLECTIVITY2"], xyz=["RANGE", "RANGE2"])  # noqa: E731
    plain = make()
    hp.decode(d_pk, plain)
    out = make()
    out, rep1 = hp.refine_placement(d_pk, out, draws=2, launches=2, ballast_gb=0.0)
    pk2, out, rep = hp.pick_placement(d_pk, make, tries=2, launches=2, stride_gb=0.0, incumbent=out, slab=False)
    assert "incumbent_ms" in rep and len(rep["output_sets_ms"]) == 2
    out, rep2 = hp.refine_placement(pk2, out, draws=2, launches=2, ballast_gb=0.0)
    assert list(out) == list(plain)
    for t in out.values():
        t.view(torch.uint8).fill_(0x3C)
    hp.decode(pk2, out)
    hp.sync()
    for k in plain:
        assert out[k].dtype == plain[k].dtype and out[k].shape == plain[k].shape, k
        assert torch.equal(out[k].view(torch.uint8), plain[k].view(torch.uint8)), k


@pytest.mark.parametrize("label,wide", VARIANTS)
@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_poses_fused_behind_the_cartesian(oracle, label, wide, dt):
    """ouster_hip_frame_out::xyz_poses: the xyz outputs are dewarp<T>(cartesian(range), poses)
    (pose_util.h:38-56) -- per-column pose, cast to T, applied while the point is in registers -- for every
    kernel variant, on clean frames, frames with holes and frames that go through the fix-up pass; everything
    else the call produces is untouched."""
    O = oracle
    cal = O.synthetic_calib(h=64, w=1024, profile="RNG15_RFL8_NIR8_DUAL")
    packets, src = O.synth_packets(cal, 4, with_window=True)
    n = 8
    pk = np.concatenate([packets, packets])
    pk[2, 5] = 0                                   # a hole: columns 80..95 of frame 2 are absent
    pk[5, [3, 9]] = pk[5, [9, 3]]                  # swapped packets: frame 5 needs the fix-up pass
    hp = _hotpath(cal, "RNG15_RFL8_NIR8_DUAL", wide=wide)
    d_pk = torch.from_numpy(pk).cuda()
    tdt = torch.float32 if dt == "f32" else torch.float64
    rng = np.random.default_rng(3)
    poses = np.tile(np.eye(4), (n, cal.w, 1, 1))
    ang = rng.uniform(-0.4, 0.4, size=(n, cal.w))
    poses[..., 0, 0] = np.cos(ang); poses[..., 0, 1] = -np.sin(ang)
    poses[..., 1, 0] = np.sin(ang); poses[..., 1, 1] = np.cos(ang)
    poses[..., :3, 3] = rng.uniform(-30, 30, size=(n, cal.w, 3))
    d_poses = torch.from_numpy(poses).cuda()
    plain = hp.alloc_outputs(n, destagger=["RANGE"], xyz=["RANGE", "RANGE2"], xyz_dtype=tdt)
    fused = hp.alloc_outputs(n, destagger=["RANGE"], xyz=["RANGE", "RANGE2"], xyz_dtype=tdt)
    hp.decode(d_pk, plain)
    hp.decode(d_pk, fused, poses=d_poses)
    hp.sync()
    for k in plain:
        if not k.startswith("xyz:") and k != "frame_meta":
            assert torch.equal(plain[k].view(torch.uint8), fused[k].view(torch.uint8)), k
    tol = 1e-4 if dt == "f32" else 1e-9      # f32: fma contraction vs separate mul / add = a few ulp at |x| ~ 300 m (1 ulp = 3e-5)
    for name in ("RANGE", "RANGE2"):
        body = _np(plain["xyz:" + name])
        got = _np(fused["xyz:" + name])
        for f in range(n):
            want = O.dewarp(body[f], poses[f], cal.h, cal.w)
            err = np.abs(got[f].astype(np.float64) - want.astype(np.float64)).max()
            assert err <= tol, (name, f, err)
        # the standalone pair gives the same cloud
        two = _np(hp.dewarp(plain["xyz:" + name], d_poses))
        assert np.abs(got.astype(np.float64) - two.astype(np.float64)).max() <= tol
    # a zero range maps to the pose's translation, like dewarp() of the (0, 0, 0) point
    r0 = _np(plain["RANGE"])
    f, r, c = np.argwhere(r0 == 0)[0]
    assert np.allclose(_np(fused["xyz:RANGE"])[f, r * cal.w + c], poses[f, c, :3, 3], atol=1e-6)


# ---------------------------------------------------------------------------------------------
# k_decode_stream: persistent workgroups, tiles double-buffered through LDS-DMA
# ---------------------------------------------------------------------------------------------
STATIC_PROFILES = ["RNG15_RFL8_NIR8_DUAL", "RNG19_RFL8_SIG16_NIR16", "RNG15_RFL8_NIR8", "LEGACY",
                   "RNG19_RFL8_SIG16_NIR16_DUAL"]


@pytest.mark.parametrize("loader", [2, 4, 0])
@pytest.mark.parametrize("rows", [0, 16])
@pytest.mark.parametrize("tw", [128, 256])
@pytest.mark.parametrize("profile", STATIC_PROFILES)
def test_stream_kernel_equals_the_one_tile_kernels(oracle, profile, tw, rows, loader):
    """Every static profile through k_decode_stream (default tile height and 8-row tiles = long per-workgroup
    pipelines): all planes, destaggered planes, xyz, column headers, packet-level outputs and the frame meta are
    byte-identical to k_decode's, on clean frames, holes, invalid columns and frames the fix-up pass redoes;
    sampled frames against the oracle."""
    O = oracle
    cal = O.synthetic_calib(h=128, w=1024, profile=profile)
    pf = cal.packet_format()
    n = 20
    packets, src = O.synth_packets(cal, 4, with_window=True)
    pad = 16 if profile == "LEGACY" else 0       # LEGACY has no packet footer: the last cell needs 16 B behind the last column
    slots = cal.w // cal.cpp
    host = np.zeros((n, slots, pf.lidar_packet_size + pad), np.uint8)
    for f in range(n):
        host[f, :, :pf.lidar_packet_size] = packets[f % 4]
    host[2, [5, 6]] = 0                                           # holes
    host[7, 0] = 0
    host[13] = 0                                                  # nothing arrived
    st_off = pf.packet_header_size + 10 if profile != "LEGACY" else pf.packet_header_size + pf.col_size - 4
    for c in (1, 7, 15):
        host[4, 9, st_off + c * pf.col_size] &= 0xFE              # invalid columns
    sw = host[9].copy(); sw[[10, 11]] = sw[[11, 10]]; host[9] = sw   # strays: the fix-up pass redoes frame 9
    host[17, 3] = host[17, 40]                                    # a duplicate in a foreign slot
    dev = torch.from_numpy(host).cuda()
    hts = torch.from_numpy((np.arange(n * slots, dtype=np.uint64) + 1000).reshape(n, slots)).cuda()
    results = {}
    names = None
    for variant in ("narrow", "stream"):
        hp = _hotpath(cal, profile, wide=0 if variant == "narrow" else "s%d" % tw, use_extrinsics=True)
        if variant == "stream":
            hp.ctx.set_knob("stream_loader", loader)   # 1: k_decode_stream2 (a ninth wave fetches), 0: every wave fetches
            if rows:
                hp.ctx.set_knob("stream_rows", rows)
        names = [x for x, _ in hp.fields]
        xyz = [x for x in ("RANGE", "RANGE2") if x in names]
        dst = [x for x in ("RANGE", "REFLECTIVITY", "NEAR_IR") if x in names]
        out = hp.alloc_outputs(n, destagger=dst, xyz=xyz)
        out["packet_timestamp"] = torch.empty((n, slots), dtype=torch.uint64, device="cuda")
        out["alert_flags"] = torch.empty((n, slots), dtype=torch.uint8, device="cuda")
        for t in out.values():
            t.view(torch.uint8).fill_(0x5C)
        hp.decode(dev, out, host_timestamps=hts)
        hp.sync()
        if variant == "stream":
            _assert_variant_ran(hp, "s%d" % tw)
            if rows:
                assert hp.ctx.last_decode_tile()[1] == rows
        results[variant] = {k: v.cpu().numpy().copy() for k, v in out.items()}
    for k, v in results["narrow"].items():
        if k == "alert_flags":     # not written for packets that did not arrive (lidar_frame.cpp:1534-1539): compare where they did
            got = results["stream"][k]
            arrived = results["narrow"]["packet_timestamp"] != 0
            assert np.array_equal(v[arrived], got[arrived]), (k, profile, tw, rows)
            continue
        assert np.array_equal(v.view(np.uint8), results["stream"][k].view(np.uint8)), (k, profile, tw, rows)
    ldir, lofs = cal.xyz_lut(True)
    for f in (0, 5, 19):
        fr = src[f % 4]
        for name in names:
            assert np.array_equal(results["stream"][name][f], fr.plane(name)), (f, name)
        for name in xyz:
            want = O.cartesian(fr.plane(name), ldir, lofs)
            assert np.abs(results["stream"]["xyz:" + name][f].astype(np.float64) - want).max() <= 1e-4


def test_tuner_times_the_persistent_kernel_too(oracle):
    """256 frames of the metric configuration with the library's own choice: over the first calls the variant tuner
    runs every candidate -- the persistent kernel among them -- and every call leaves the same bytes."""
    O = oracle
    cal = O.synthetic_calib(h=128, w=2048, profile="RNG15_RFL8_NIR8_DUAL")
    packets, _ = O.synth_packets(cal, 8, with_window=True)
    d_pk = torch.from_numpy(packets).cuda().repeat(32, 1, 1).contiguous()
    hp = _hotpath(cal, "RNG15_RFL8_NIR8_DUAL", wide=None)
    out = hp.alloc_outputs(256, destagger=["RANGE", "RANGE2"], xyz=["RANGE", "RANGE2"])
    seen, first = set(), None
    for call in range(24):
        for t in out.values():
            t.view(torch.uint8).fill_(0x77)
        hp.decode(d_pk, out)
        hp.sync()
        seen.add((hp.ctx.last_decode_kernel(),) + tuple(hp.ctx.last_decode_tile()))
        snap = {k: v.clone() for k, v in out.items()}
        if first is None:
            first = snap
        for k in first:
            assert torch.equal(first[k].view(torch.uint8), snap[k].view(torch.uint8)), (call, k, seen)
    kernels = {k for k, _, _ in seen}
    assert {"k_decode", "k_decode_wide"} <= kernels and kernels & {"k_decode_stream", "k_decode_stream2"}, seen


# ---------------------------------------------------------------------------------------------
# buffers WITHOUT one slot per column of the frame: the general mapping, resolved once per frame by k_slotmap and
# decoded on k_decode_wide's tiles (or, slotmap = 0 / narrow forced, by every 64-column tile of k_decode for itself)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("wide", [None, 64, 128, 256, 0])
@pytest.mark.parametrize("profile,slots", [("RNG15_RFL8_NIR8_DUAL", 67), ("RNG19_RFL8_SIG16_NIR16", 61),
                                           ("RNG15_RFL8_NIR8", 70), ("FUSA_RNG15_RFL8_NIR8_DUAL", 64 + 5)])
def test_general_mapping_on_wide_tiles(oracle, profile, slots, wide):
    """parse_by_col semantics (lidar_frame.cpp:1422-1466) for every frame of a batch whose buffer has `slots` != W / cpp
    packet slots: compacted after drops, any order, a late duplicate that must win, an empty frame, a full frame, packets
    whose columns lie outside the frame.  Everything -- planes, destaggered planes, XYZ, column headers, packet-level
    outputs, frame meta -- against the oracle fed the same packets in the same order."""
    O = oracle
    cal = O.synthetic_calib(h=32, w=1024, profile=profile)
    pf = cal.packet_format()
    n, npk = 12, 64
    packets, _ = O.synth_packets(cal, n, with_window=True)
    rng = np.random.default_rng(5)
    by_frame = [packets[f] for f in range(n)]
    by_frame[1] = np.delete(by_frame[1], [0, 9, 63], axis=0)
    by_frame[2] = by_frame[2][rng.permutation(npk)]
    dup = by_frame[3][30:31].copy()
    body = pf.packet_header_size + pf.col_header_size
    dup[0, body:body + 64] ^= 0x15   # the late copy differs: it is the one that must be decoded
    by_frame[3] = np.concatenate([by_frame[3], dup])[: min(slots, npk + 1)]
    by_frame[4] = by_frame[4][:0]
    by_frame[5] = np.delete(by_frame[5], np.arange(1, npk, 2), axis=0)[rng.permutation(npk // 2)]
    mid_ofs = pf.col_measurement_id_info.offset
    assert pf.col_measurement_id_info.mask == 0xffff and pf.col_measurement_id_info.shift == 0
    far = by_frame[6][:2].copy()   # measurement ids beyond the frame: dropped (lidar_frame.cpp:1432-1434)
    for k in range(2):
        for ic in range(cal.cpp):
            o = pf.packet_header_size + ic * pf.col_size + mid_ofs
            far[k, o:o + 2] = np.frombuffer(np.uint16(cal.w + 16 * k + ic).tobytes(), np.uint8)
    by_frame[6] = np.concatenate([np.delete(by_frame[6], [5], axis=0), far])[:slots]
    by_frame[7] = by_frame[7][::-1].copy()
    by_frame = [b[:slots] for b in by_frame]
    host = np.zeros((n, slots, pf.lidar_packet_size), np.uint8)
    counts = np.zeros(n, np.uint32)
    for f, pk in enumerate(by_frame):
        host[f, :len(pk)] = pk
        counts[f] = len(pk)
    names = [nm for nm, _ in HotPath(profile, cal.h, cal.w, cal.cpp, header_type=cal.header_type).fields]
    xyz = [nm for nm in ("RANGE", "RANGE2") if nm in names]
    dst = [nm for nm in ("RANGE", "REFLECTIVITY", "RANGE2") if nm in names]
    ref = _oracle_frames(O, cal, pf, by_frame, True)
    hp = _hotpath(cal, profile, wide=wide)
    if wide is None:
        hp.ctx.set_knob("wide_min_blocks", 0)
    out = hp.alloc_outputs(n, destagger=dst, xyz=xyz)
    out["packet_timestamp"] = torch.empty((n, npk), dtype=torch.uint64, device="cuda")
    out["alert_flags"] = torch.empty((n, npk), dtype=torch.uint8, device="cuda")
    for t in out.values():
        t.view(torch.uint8).fill_(0xCD)
    ts = torch.arange(1, n * slots + 1, dtype=torch.int64).view(n, slots).cuda() * 1000
    hp.decode(torch.from_numpy(host).cuda(), out, packet_counts=counts, host_timestamps=ts)
    hp.sync()
    kernel = hp.ctx.last_decode_kernel()
    assert kernel == ("k_decode" if wide == 0 else "k_decode_wide"), kernel
    _compare(O, cal, hp, out, ref, dst, xyz)
    # packet-level outputs: the LAST buffered packet of each packet index (batch_lidar_packet, lidar_frame.cpp:1534-1539)
    pts = _np(out["packet_timestamp"])
    for f, pk in enumerate(by_frame):
        want = np.zeros(npk, np.uint64)
        for sl, p in enumerate(pk):
            o = pf.packet_header_size + mid_ofs
            idx = int(np.frombuffer(p[o:o + 2].tobytes(), np.uint16)[0]) // cal.cpp
            if idx < npk:
                want[idx] = (f * slots + sl + 1) * 1000
        assert np.array_equal(pts[f], want), f
    # alert_flags: written for the packets that arrived, untouched elsewhere (not zeroed at frame start, lidar_frame.cpp:1719)
    al = _np(out["alert_flags"])
    for f, pk in enumerate(by_frame):
        assert np.all(al[f][pts[f] == 0] == 0xCD), f


# ---------------------------------------------------------------------------------------------
# the loss paths bench.py times next to the metric ("loss_paths"), at the benchmarked size
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", ["stray10", "general", "general_narrow"])
def test_bench_size_loss_paths(oracle, path):
    """256 frames of 128x2048 dual return with bench.py's loss patterns: `stray10` (every 20th frame compacted after a
    drop, every 20th + 10 with two packets swapped: fix-up pass) and `general` (every frame compacted into 127 slots with
    per-frame packet counts: MODE_GENERAL for every frame, the reference's parse_by_col fallback,
    lidar_frame.cpp:1422-1466, packet_format_test.cpp:328-406).  Lossy and clean frames against the oracle, all frames
    against the period of the inputs."""
    O = oracle
    profile = "RNG15_RFL8_NIR8_DUAL"
    cal = O.synthetic_calib(h=128, w=2048, profile=profile)
    pf = cal.packet_format()
    packets, src = O.synth_packets(cal, 8, with_window=True)
    n, slots = 256, 128
    hp = _hotpath(cal, profile)
    dst, xyz = ["RANGE", "RANGE2", "REFLECTIVITY", "REFLECTIVITY2"], ["RANGE", "RANGE2"]
    lost = [(f * 7 + 3) % slots for f in range(n)]
    by_frame = {}
    if path == "general_narrow":
        hp.ctx.set_knob("slotmap", 0)
    if path.startswith("general"):
        host = np.zeros((n, slots - 1, pf.lidar_packet_size), np.uint8)
        for f in range(n):
            host[f] = np.delete(packets[f % 8], lost[f], axis=0)
        counts = np.full(n, slots - 1, np.uint32)
        check = [0, 7, 100, 255]
        for f in check:
            by_frame[f] = host[f]
    else:
        host = np.zeros((n, slots, pf.lidar_packet_size), np.uint8)
        counts = np.full(n, slots, np.uint32)
        for f in range(n):
            host[f] = packets[f % 8]
            if f % 20 == 3:
                host[f, :slots - 1] = np.delete(packets[f % 8], lost[f], axis=0)
                host[f, slots - 1] = 0
                counts[f] = slots - 1
            elif f % 20 == 13:
                host[f, [10, 11]] = host[f, [11, 10]]
        check = [0, 3, 13, 23, 243, 253, 255]
        for f in check:
            by_frame[f] = host[f, :counts[f]]
    out = hp.alloc_outputs(n, destagger=dst, xyz=xyz)
    for t in out.values():
        t.view(torch.uint8).fill_(0xA5)
    hp.decode(torch.from_numpy(host).cuda(), out, packet_counts=counts)
    hp.sync()
    if path.startswith("general"):
        assert hp.ctx.last_decode_kernel() == ("k_decode" if path == "general_narrow" else "k_decode_wide")
    ref = _oracle_frames(O, cal, pf, [by_frame[f] for f in check], True)
    worst = _compare(O, cal, hp, out, ref, dst, xyz, frames=list(zip(check, ref)))
    assert worst <= 4e-5


@pytest.mark.parametrize("wide", [256, 128])
@pytest.mark.parametrize("misalign", [4, 8, 12])
def test_wide_tiles_on_a_packet_buffer_that_is_not_16_byte_aligned(oracle, wide, misalign):
    """The wide kernel stages 16-byte chunks aligned to ABSOLUTE addresses.  When the packet buffer starts 4 / 8 / 12 bytes
    off a 16-byte boundary and is packed tightly (LEGACY: no packet footer), the last chunk of the last packet of a frame
    would reach past the frame's bytes: it is read as the 16 bytes that END there and its dwords are shifted into place.
    All frames -- the last one ends at the last byte of the allocation -- equal the oracle."""
    O = oracle
    profile = "LEGACY"
    cal = O.synthetic_calib(h=128, w=1024, profile=profile)
    pf = cal.packet_format()
    n = 3
    packets, src = O.synth_packets(cal, n)
    slots = cal.w // cal.cpp
    total = n * slots * pf.lidar_packet_size
    flat = torch.zeros(total + 16, dtype=torch.uint8, device="cuda")
    assert flat.data_ptr() % 16 == 0
    flat[misalign:misalign + total] = torch.from_numpy(np.ascontiguousarray(packets).reshape(-1)).cuda()
    dev = flat[misalign:misalign + total].view(n, slots, pf.lidar_packet_size)
    assert dev.data_ptr() % 16 == misalign
    hp = _hotpath(cal, profile, wide=wide, use_extrinsics=False)
    names = [x for x, _ in hp.fields]
    out = hp.alloc_outputs(n, destagger=["RANGE"], xyz=["RANGE"])
    for t in out.values():
        t.view(torch.uint8).fill_(0x3B)
    hp.decode(dev, out)
    hp.sync()
    _assert_variant_ran(hp, wide)
    _compare(O, cal, hp, out, src, ["RANGE"], ["RANGE"])
