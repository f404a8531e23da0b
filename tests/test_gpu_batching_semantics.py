"""What FrameBatcher leaves behind for packets no sensor sends -- the two deviations DESIGN.md section 5 used to list
 and a fuzz over everything in between, GPU against the oracle's sequential
restatement of FrameBatcher::batch (lidar_frame.cpp:1422-1576), on every decode kernel variant and through both routes:
one slot per column of the frame (optimistic pass + fix-up pass) and a compacted buffer with packet counts (general mapping).

  * block path: a packet whose columns are all valid and in range is parsed by block whether or not its measurement ids are
    consecutive: headers go to every column's own id, PIXELS to m_id(first column of the block) + x
    (parse_by_block :1492-1528, block_field parsing.cpp:628-654);
  * a re-sent packet whose copies disagree about which columns are valid: batched column by column, each copy's valid
    columns land, the later copy wins where both are valid (parse_by_col :1422-1466);
  * next_valid_m_id bookkeeping: what a jump forward zeroes (headers too) and what the end of the frame zeroes (planes only).
"""
import numpy as np
import pytest

from conftest import has_gpu

pytestmark = pytest.mark.gpu

if has_gpu():
    import torch

from test_gpu_parity import _np  # noqa: E402
from test_gpu_fastpath import VARIANTS, _hotpath, _assert_variant_ran  # noqa: E402

PROFILE = "RNG15_RFL8_NIR8_DUAL"


def _set_mid(pf, pkt, ic, m_id):
    o = pf.packet_header_size + ic * pf.col_size + 8
    pkt[o:o + 2] = np.frombuffer(np.uint16(m_id).tobytes(), np.uint8)


def _set_valid(pf, pkt, ic, valid):
    o = pf.packet_header_size + ic * pf.col_size + 10
    pkt[o] = (pkt[o] | 1) if valid else (pkt[o] & 0xFE)


def _reference_frames(O, cal, pf, by_frame, hts):
    """The oracle batcher over each frame's packets in buffer order, host timestamps as given."""
    frames = []
    for f, pk in enumerate(by_frame):
        fr = O.Frame.for_profile(cal.profile, cal.h, cal.w, cal.cpp, with_window=True)
        fr.fill(0)   # planes the reference neither writes nor zeroes keep the frame's previous contents: zeros here
        if len(pk):
            b = O.Batcher(pf, init_id=O.lib().ora_init_id(pf, pk[0].ctypes.data), expected_packets=len(pk))
            done = False
            for i, p in enumerate(pk):
                done = b.batch(p, int(hts[f][i]), fr)
            if not done:
                b.finalize(fr)
        frames.append(fr)
    return frames


def _decode_and_compare(O, cal, pf, by_frame, wide, compacted, label=""):
    """compacted=False: every frame in W / cpp slots (missing packets = zeroed slots at the END of the buffer order the test
    chose); compacted=True: one slot fewer or more than W / cpp, packet counts given: the general mapping for every frame."""
    n, ppf = len(by_frame), cal.w // cal.cpp
    longest = max(max(len(p) for p in by_frame), 1)
    slots = ppf if not compacted else (longest if longest != ppf else ppf + 1)
    assert all(len(p) <= slots for p in by_frame)
    host = np.zeros((n, slots, pf.lidar_packet_size), np.uint8)
    counts = np.zeros(n, np.uint32)
    hts = np.zeros((n, slots), np.uint64)
    for f, pk in enumerate(by_frame):
        host[f, :len(pk)] = pk
        counts[f] = len(pk)
        hts[f] = 1000 * (f + 1) + np.arange(slots)
    hp = _hotpath(cal, PROFILE, wide=wide)
    dst, xyz = ["RANGE", "REFLECTIVITY2"], ["RANGE", "RANGE2"]
    out = hp.alloc_outputs(n, destagger=dst, xyz=xyz)
    for t in out.values():
        t.view(torch.uint8).fill_(0xCD)
    out["packet_timestamp"] = torch.full((n, ppf), 7, dtype=torch.int64, device="cuda").to(torch.uint64)
    out["alert_flags"] = torch.zeros((n, ppf), dtype=torch.uint8, device="cuda")
    hp.decode(torch.from_numpy(host).cuda(), out, packet_counts=counts, host_timestamps=torch.from_numpy(hts).cuda())
    hp.sync()
    ref = _reference_frames(O, cal, pf, by_frame, hts)
    ldir, lofs = cal.xyz_lut(False)
    for f, fr in enumerate(ref):
        tag = (label, f, wide, compacted)
        for name, _ in hp.fields:
            assert np.array_equal(_np(out[name][f]), fr.plane(name)), (name,) + tag
        assert np.array_equal(_np(out["measurement_id"][f]), fr.measurement_id), tag
        assert np.array_equal(_np(out["status"][f]), fr.status), tag
        assert np.array_equal(_np(out["timestamp"][f]), fr.timestamp), tag
        assert np.array_equal(_np(out["packet_timestamp"][f]), fr.packet_timestamp), tag
        assert np.array_equal(_np(out["alert_flags"][f]), fr.alert_flags), tag
        for name in dst:
            assert np.array_equal(_np(out["destaggered:" + name][f]), O.destagger(fr.plane(name), cal.pixel_shift_by_row)), (name,) + tag
        for name in xyz:
            want = O.cartesian(fr.plane(name), ldir, lofs)
            assert np.abs(_np(out["xyz:" + name][f]).astype(np.float64) - want).max() <= 1e-4, (name,) + tag
        meta = _np(out["frame_meta"][f])
        assert meta[20:24].view(np.uint32)[0] == int((fr.status & 1).sum()), ("n_valid_columns",) + tag
    return hp


@pytest.mark.parametrize("compacted", [False, True])
@pytest.mark.parametrize("label,wide", VARIANTS)
def test_all_valid_packet_with_non_consecutive_ids_follows_the_block_path(oracle, label, wide, compacted):
    """VERDICT r03 item 2.  Frame 1: one packet's ids shuffled among themselves; frame 2: a packet whose first column
    claims another packet's place (pixels follow the first id, headers their own ids); frame 3: ids scattered over the
    frame; frame 4: the same packet, one column invalid -> the reference falls back to the column path and every pixel goes
    to its own id.  Frames 0 and 5 are clean."""
    O = oracle
    cal = O.synthetic_calib(h=64, w=1024, profile=PROFILE)
    pf = cal.packet_format()
    packets, _ = O.synth_packets(cal, 6, with_window=True)
    rng = np.random.default_rng(21)
    by_frame = [packets[f].copy() for f in range(6)]
    perm = rng.permutation(16)
    for ic in range(16):
        _set_mid(pf, by_frame[1][9], ic, 9 * 16 + int(perm[ic]))
    for ic in range(16):
        _set_mid(pf, by_frame[2][20], ic, 33 * 16 + ic if ic < 4 else 20 * 16 + ic)
    scattered = rng.choice(1024 - 16, 16, replace=False)
    for f in (3, 4):
        for ic in range(16):
            _set_mid(pf, by_frame[f][40], ic, int(scattered[ic]))
    _set_valid(pf, by_frame[4][40], 5, False)
    if compacted:
        by_frame = [np.delete(b, [3], axis=0) for b in by_frame]
    hp = _decode_and_compare(O, cal, pf, by_frame, wide, compacted, "block path")
    if not compacted:
        _assert_variant_ran(hp, wide)


@pytest.mark.parametrize("label,wide", VARIANTS)
def test_resent_packet_whose_copies_disagree_about_valid_columns(oracle, label, wide):
    """VERDICT r03 item 3.  Packet 12 arrives twice; the first copy has columns 0..7 valid, the second columns 4..11: the
    frame ends up with 0..3 from the first copy, 4..11 from the second, 12..15 zero.  One slot per column (the second copy
    sits in the slot of a packet that was lost: a stray, fix-up pass) and as an extra slot (general mapping)."""
    O = oracle
    cal = O.synthetic_calib(h=64, w=1024, profile=PROFILE)
    pf = cal.packet_format()
    packets, _ = O.synth_packets(cal, 3, with_window=True)
    other, _ = O.synth_packets(cal, 3, seed=99, with_window=True)
    first, second = packets[1][12].copy(), other[1][12].copy()
    second[:pf.packet_header_size] = first[:pf.packet_header_size]
    for ic in range(16):
        _set_valid(pf, first, ic, ic < 8)
        _set_valid(pf, second, ic, 4 <= ic < 12)
    # one slot per column: packet 30 never arrived, the second copy sits in its slot
    a = packets[1].copy()
    a[12] = first
    a[30] = second
    _decode_and_compare(O, cal, pf, [packets[0], a, packets[2]], wide, False, "resent, in a lost packet's slot")
    # an extra slot at the end of the buffer
    b = np.concatenate([packets[1][:12], first[None], packets[1][13:], second[None]])
    _decode_and_compare(O, cal, pf, [packets[0], b, packets[2]], wide, True, "resent, extra slot")
    # and the later copy in an EARLIER slot than the first (buffer order decides, not the packet id)
    c = packets[1].copy()
    c[12] = first
    c[5] = second
    _decode_and_compare(O, cal, pf, [packets[0], c, packets[2]], wide, False, "resent, earlier slot")


def _fuzz_frame(pf, rng, pk, w, cpp):
    """One frame's packets after a random sequence of the things a network / a confused sender can do."""
    ppf = len(pk)
    pk = pk.copy()
    order = list(range(ppf))
    kind = rng.integers(0, 8)
    if kind in (1, 5, 7):
        rng.shuffle(order)
    if kind in (2, 5):
        for p in rng.choice(ppf, rng.integers(1, 6), replace=False):
            order.remove(int(p))
    out = [pk[p].copy() for p in order]
    if kind in (3, 5, 7):                          # duplicates, some with other valid bits
        for _ in range(rng.integers(1, 4)):
            src = out[int(rng.integers(0, len(out)))].copy()
            if rng.random() < 0.5:
                for ic in rng.choice(cpp, 5, replace=False):
                    _set_valid(pf, src, int(ic), False)
            out.insert(int(rng.integers(0, len(out) + 1)), src)
    if kind in (4, 6, 7):                          # ids rewritten: shuffled inside a packet, moved, out of range, invalid
        for _ in range(rng.integers(1, 5)):
            p = out[int(rng.integers(0, len(out)))]
            how = rng.integers(0, 5)
            if how == 0:
                base = int(np.frombuffer(p[pf.packet_header_size + 8:pf.packet_header_size + 10].tobytes(), np.uint16)[0])
                for ic, m in enumerate(rng.permutation(cpp)):
                    _set_mid(pf, p, ic, base - base % cpp + int(m))
            elif how == 1:
                for ic in range(cpp):
                    _set_mid(pf, p, ic, int(rng.integers(0, w)))
            elif how == 2:
                _set_mid(pf, p, int(rng.integers(0, cpp)), int(rng.integers(w, w + 300)))
            elif how == 3:
                _set_mid(pf, p, 0, int(rng.integers(0, w - cpp)))
            else:
                for ic in rng.choice(cpp, int(rng.integers(1, cpp)), replace=False):
                    _set_valid(pf, p, int(ic), False)
    return np.stack(out) if out else np.zeros((0, pk.shape[1]), np.uint8)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("label,wide", [VARIANTS[0], VARIANTS[3], VARIANTS[5]])
def test_fuzz_against_the_sequential_reference(oracle, label, wide, seed):
    """Random orders, drops, duplicates with other valid bits, rewritten ids -- 12 frames per case, decoded through whichever
    route the buffer shape selects: frames that fit W / cpp slots exactly go through the optimistic pass + fix-up pass, the
    rest (as one batch with a longer buffer) through the general mapping."""
    O = oracle
    cal = O.synthetic_calib(h=32, w=512, profile=PROFILE)
    pf = cal.packet_format()
    packets, _ = O.synth_packets(cal, 12, seed=100 + seed, with_window=True)
    rng = np.random.default_rng(1000 + seed)
    frames = [_fuzz_frame(pf, rng, packets[f], cal.w, cal.cpp) for f in range(12)]
    ppf = cal.w // cal.cpp
    fits = [fr for fr in frames if len(fr) <= ppf]
    if fits:
        _decode_and_compare(O, cal, pf, fits, wide, False, f"fuzz {seed} slots")
    _decode_and_compare(O, cal, pf, frames, wide, True, f"fuzz {seed} compacted")


def _damaged_batch(O, cal, n, seed):
    """n frames, every fifth one damaged in turn: two packets swapped / compacted after a drop / a duplicate in a lost packet's
    slot / one packet's ids shuffled."""
    pf = cal.packet_format()
    packets, _ = O.synth_packets(cal, n, seed=seed, with_window=True)
    rng = np.random.default_rng(seed)
    ppf = cal.w // cal.cpp
    by_frame = []
    for f in range(n):
        pk = packets[f].copy()
        kind = (f // 5) % 4 if f % 5 == 2 else -1
        if kind == 0:
            a, b = rng.choice(ppf, 2, replace=False)
            pk[[a, b]] = pk[[b, a]]
        elif kind == 1:
            pk = np.delete(pk, int(rng.integers(0, ppf - 1)), axis=0)
        elif kind == 2:
            lost, src = rng.choice(ppf, 2, replace=False)
            pk[lost] = pk[src]
        elif kind == 3:
            p = int(rng.integers(0, ppf))
            for ic, m in enumerate(rng.permutation(cal.cpp)):
                _set_mid(pf, pk[p], ic, p * cal.cpp + int(m))
        by_frame.append(pk)
    return pf, by_frame


def test_more_frames_than_one_fixup_chunk(oracle):
    """The fix-up pass lists flagged frames 512 at a time; tickets run on across the chunks.  700 small frames, damaged ones in
    both chunks (and in the last, partial one): more tickets than workgroups, so both sources of a REDO ticket's maps are
    exercised -- resolved by the workgroup itself (first round) and read from what a LEAD ticket published (later rounds)."""
    O = oracle
    cal = O.synthetic_calib(h=16, w=256, profile=PROFILE)
    pf, by_frame = _damaged_batch(O, cal, 700, 11)
    hp = _decode_and_compare(O, cal, pf, by_frame, None, False, "700 frames")
    assert hp.ctx.last_decode_kernel() != ""


@pytest.mark.parametrize("h,w,cpp", [(64, 4096, 16), (32, 512, 8), (16, 384, 12), (20, 512, 16), (16, 256, 4)])
def test_other_geometries_take_the_same_semantics(oracle, h, w, cpp):
    """4096 columns (the column maps no longer fit under a wide tile: the fix-up pass falls back to 64-column tiles, each
    resolving its frame), 8 and 4 columns per packet (other lane groups), 12 (not a power of two: the per-packet fallback of
    resolve_frame), 20 rows (block_parsable() = 4 < columns_per_packet)."""
    O = oracle
    cal = O.synthetic_calib(h=h, w=w, cpp=cpp, profile=PROFILE)
    pf = cal.packet_format()
    packets, _ = O.synth_packets(cal, 6, seed=h + w + cpp, with_window=True)
    ppf = w // cpp
    rng = np.random.default_rng(h * w + cpp)
    by_frame = [packets[f].copy() for f in range(6)]
    a, b = rng.choice(ppf, 2, replace=False)
    by_frame[1][[a, b]] = by_frame[1][[b, a]]                                        # two packets swapped
    by_frame[2] = np.delete(by_frame[2], int(rng.integers(0, ppf - 1)), axis=0)       # compacted after a drop
    p = int(rng.integers(0, ppf))
    for ic, m in enumerate(rng.permutation(cpp)):                                     # ids shuffled inside an all-valid packet
        _set_mid(pf, by_frame[3][p], ic, p * cpp + int(m))
    q = int(rng.integers(1, ppf))
    first = by_frame[4][q].copy()
    for ic in range(cpp // 2, cpp):
        _set_valid(pf, first, ic, False)                                             # a first copy that ends in invalid columns ...
    by_frame[4] = np.concatenate([by_frame[4][:q], first[None], by_frame[4][q:]])[:ppf + 1]   # ... then the packet again, whole
    # the same through the fix-up pass: the whole copy sits in the slot of the (lost) next packet.  The reference leaves
    # next_valid inside the packet's span after the first copy, writes the second without moving it, and the next jump
    # zeroes the second copy's tail again -- reproduced, not repaired
    r = int(rng.integers(0, ppf - 2))
    whole = by_frame[5][r].copy()
    part = whole.copy()
    for ic in range(cpp // 2, cpp):
        _set_valid(pf, part, ic, False)
    by_frame[5][r], by_frame[5][r + 1] = part, whole
    for compacted in (False, True):
        frames = by_frame if compacted else [fr for fr in by_frame if len(fr) <= ppf]
        _decode_and_compare(O, cal, pf, frames, None, compacted, f"{h}x{w} cpp {cpp}")
