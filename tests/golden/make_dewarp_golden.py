#!/usr/bin/env python3
"""Generates tests/golden/dewarp_ref_vectors.npz: inputs and the outputs of the REFERENCE's own impl/dewarp_impl.h
(oracle/_ref/libdewarp_ref.so, built by oracle/Makefile where /root/reference exists) for a small FrameSet: three
frames with invalid columns / zero-status columns inside the valid span, one frame without any valid column, one
absent frame.  tests/test_oracle_ref_dewarp.py replays them against the oracle where the library is absent.
Run from the repo root: python tests/golden/make_dewarp_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dewarp_ref  # noqa: E402
from oracle import oracle as O  # noqa: E402


def inputs():
    O.build()
    h, w, n = 16, 64, 5
    cal = O.synthetic_calib(h=h, w=w, cpp=16)
    d, o = cal.xyz_lut(True)
    g = np.random.default_rng(2026)
    r = g.integers(0, 60000, (n, h, w)).astype(np.uint32)
    r[g.random(r.shape) < 0.25] = 0
    st = np.ones((n, w), np.uint32)
    st[0, :3] = 0; st[0, 10] = 0; st[0, 11] = 2; st[0, 60:] = 0       # zero / even status inside the valid span
    st[1, ::2] = 0                                                  # every other column missing
    st[2] = 0                                                       # no valid column: empty contribution
    st[3, 5:9] = 6                                                  # status without bit 0: emitted, but never first / last
    ts = (np.arange(n * w).reshape(n, w) * 1000 + 7).astype(np.uint64)
    poses = np.tile(np.eye(4), (n, w, 1, 1))
    ang = g.uniform(-0.5, 0.5, (n, w))
    poses[..., 0, 0] = np.cos(ang); poses[..., 0, 1] = -np.sin(ang)
    poses[..., 1, 0] = np.sin(ang); poses[..., 1, 1] = np.cos(ang)
    poses[..., :3, 3] = g.uniform(-40, 40, (n, w, 3))
    present = np.array([1, 1, 1, 1, 0], np.uint8)                   # frame 4 is an empty slot of the set
    return dict(range=r, status=st, timestamp=ts, poses=poses, lut_dir=d, lut_ofs=o, present=present,
                gate=np.array([1.0, 30.0]))


def main():
    assert dewarp_ref.available(), "oracle/_ref/libdewarp_ref.so missing: make -C oracle (needs /root/reference)"
    v = inputs()
    out = dict(v)
    for tag, T in (("f64", np.float64), ("f32", np.float32)):
        p, fi, ci, tn = dewarp_ref.dewarp_frames(v["range"], v["status"], v["timestamp"], v["poses"],
                                                 v["lut_dir"].astype(T), v["lut_ofs"].astype(T), *v["gate"],
                                                 present=v["present"])
        out.update({f"points_{tag}": p, f"frame_idxs_{tag}": fi, f"col_idxs_{tag}": ci, f"timestamps_{tag}": tn})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dewarp_ref_vectors.npz"), **out)
    print({k: getattr(x, "shape", x) for k, x in out.items()})


if __name__ == "__main__":
    main()
