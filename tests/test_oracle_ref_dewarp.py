"""The oracle's range-gated frame dewarp (ora_dewarp_frame_*) pinned on the REFERENCE'S OWN code:
impl/dewarp_impl.h compiled from /root/reference into oracle/_ref/libdewarp_ref.so (oracle/Makefile, tests/dewarp_ref.py)
where the reference checkout exists, and its outputs committed as tests/golden/dewarp_ref_vectors.npz
(tests/golden/make_dewarp_golden.py) for everywhere else.  Counts, order, frame / column indices and timestamps must be
identical; points agree to the last bits (the reference sums its 3-term dot product as Eigen's unrolled reduction does,
p0 + (p1 + p2), the oracle left to right).  Also the dense dewarp's rotation KAT of the reference's Python tests."""
import os

import numpy as np
import pytest

import dewarp_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dewarp_ref_vectors.npz")


def _oracle_frames(O, v, T):
    pts, fi, ci, tn = [], [], [], []
    for f in range(v["range"].shape[0]):
        if not v["present"][f]:
            continue
        p, c, t = O.dewarp_frame(v["range"][f], v["status"][f], v["timestamp"][f], v["poses"][f],
                                 v["lut_dir"].astype(T), v["lut_ofs"].astype(T), *v["gate"])
        pts.append(p); ci.append(c); tn.append(t); fi.append(np.full(len(c), f, np.uint32))
    return np.concatenate(pts), np.concatenate(fi), np.concatenate(ci), np.concatenate(tn)


@pytest.mark.parametrize("tag,T,rtol", [("f64", np.float64, 1e-14), ("f32", np.float32, 4e-6)])
def test_oracle_equals_the_committed_reference_outputs(oracle, tag, T, rtol):
    """A FrameSet of five: invalid columns, zero / even status inside the valid span, a frame without valid columns, an
    absent frame -- the reference's outputs (committed) against the oracle's."""
    v = dict(np.load(GOLDEN))
    p, fi, ci, tn = _oracle_frames(oracle, v, T)
    assert np.array_equal(fi, v[f"frame_idxs_{tag}"])
    assert np.array_equal(ci, v[f"col_idxs_{tag}"])
    assert np.array_equal(tn, v[f"timestamps_{tag}"])
    assert p.dtype == v[f"points_{tag}"].dtype and p.shape == v[f"points_{tag}"].shape and len(p) > 500
    scale = np.abs(v[f"points_{tag}"]).max()
    assert np.abs(p.astype(np.float64) - v[f"points_{tag}"].astype(np.float64)).max() <= rtol * scale
    assert 2 not in fi and 4 not in fi            # the frame without valid columns and the absent one contribute nothing
    f0 = ci[fi == 0]
    assert f0.min() == 3 and f0.max() == 59 and 10 not in f0 and 11 in f0   # status 0 skipped, status 2 emitted
    f3 = ci[fi == 3]
    assert {5, 6, 7, 8} <= set(f3.tolist())       # status 6 (bit 0 clear) inside the span: emitted (only == 0 is skipped)


@pytest.mark.skipif(not dewarp_ref.available(), reason="oracle/_ref/libdewarp_ref.so is built only where /root/reference exists")
@pytest.mark.parametrize("T,rtol", [(np.float64, 1e-14), (np.float32, 4e-6)])
def test_oracle_equals_the_compiled_reference(oracle, T, rtol):
    """The same comparison live, on fresh random frames (single-frame entry point too), plus: the committed vectors are
    what the compiled reference produces today."""
    O = oracle
    from golden import make_dewarp_golden as G
    v = G.inputs()
    got = dewarp_ref.dewarp_frames(v["range"], v["status"], v["timestamp"], v["poses"], v["lut_dir"].astype(T),
                                   v["lut_ofs"].astype(T), *v["gate"], present=v["present"])
    tag = "f64" if T == np.float64 else "f32"
    gold = dict(np.load(GOLDEN))
    for a, k in zip(got, ("points", "frame_idxs", "col_idxs", "timestamps")):
        assert np.array_equal(a, gold[f"{k}_{tag}"]), k
    g = np.random.default_rng(77)
    h, w = 32, 128
    cal = O.synthetic_calib(h=h, w=w, cpp=16)
    d, o = cal.xyz_lut(False)
    for trial in range(4):
        r = g.integers(0, 200000, (h, w)).astype(np.uint32)
        r[g.random(r.shape) < 0.3] = 0
        st = (g.random(w) < 0.8).astype(np.uint32) * g.integers(1, 4, w).astype(np.uint32)
        if trial == 3:
            st[:] = 0
        ts = g.integers(0, 1 << 60, w).astype(np.uint64)
        poses = g.normal(size=(w, 4, 4))
        poses[:, 3] = [0, 0, 0, 1]
        lo, hi = (0.0, 1000.0) if trial == 0 else (0.5 + trial, 60.0 * trial)
        a = O.dewarp_frame(r, st, ts, poses, d.astype(T), o.astype(T), lo, hi)
        b = dewarp_ref.dewarp_frame(r, st, ts, poses, d.astype(T), o.astype(T), lo, hi)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), trial
        if len(b[0]):
            assert np.abs(a[0].astype(np.float64) - b[0].astype(np.float64)).max() <= rtol * np.abs(b[0]).max() * 4
        else:
            assert len(a[0]) == 0 and trial == 3


def test_dense_dewarp_rotation_kat(oracle):
    """python/tests/test_pose_util.py:300-331 (test_transform_N_M_3): a 30 degree yaw with translation (1, 2, -1) -- pins
    the element order of R in pt = R * p + t (the translation-only KAT of :334-359 does not); and :334-359 itself."""
    O = oracle
    pts = np.arange(1.0, 25.0).reshape(2, 4, 3)                    # (h = 2, w = 4)
    tf = np.array([[0.866, -0.5, 0.0, 1.0], [0.5, 0.866, 0.0, 2.0], [0.0, 0.0, 1.0, -1.0], [0.0, 0.0, 0.0, 1.0]])
    want = np.array([[[0.866, 4.232, 2], [1.964, 8.33, 5], [3.062, 12.428, 8], [4.16, 16.526, 11]],
                     [[5.258, 20.624, 14], [6.356, 24.722, 17], [7.454, 28.82, 20], [8.552, 32.918, 23]]])
    poses = np.tile(tf, (4, 1, 1))                                 # the same pose for every column = transform()
    got = O.dewarp(pts.reshape(8, 3), poses, 2, 4).reshape(2, 4, 3)
    np.testing.assert_almost_equal(got, want, decimal=5)
    got32 = O.dewarp(pts.reshape(8, 3).astype(np.float32), poses, 2, 4).reshape(2, 4, 3)
    np.testing.assert_almost_equal(got32, want, decimal=4)
    # test_dewarp (:334-359): per-column translation poses, points laid out (pts_per_pose, num_poses, 3)
    poses = np.array([[1, 0, 0, 1, 0, 1, 0, -2, 0, 0, 1, 3, 0, 0, 0, 1]] * 4, dtype=np.float64).reshape(4, 4, 4)
    points = np.array([[i - 3, i + 1, i + 2] for i in range(8)], dtype=np.float64)
    want = np.array([[[-2, -1, 5], [-1, 0, 6], [0, 1, 7], [1, 2, 8]], [[2, 3, 9], [3, 4, 10], [4, 5, 11], [5, 6, 12]]], np.float64)
    np.testing.assert_allclose(O.dewarp(points, poses, 2, 4).reshape(2, 4, 3), want, rtol=1e-5, atol=1e-8)
