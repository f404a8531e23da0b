#!/usr/bin/env python3
"""Writes tests/golden/osf/zpng_ref_vectors.json: field planes compressed by the REFERENCE's own ZPNG codec
(oracle/_ref/libzpng_ref.so = /root/reference/thirdparty/zpng/zpng.cpp, built by oracle/Makefile) in the
layouts the reference's OSF writer uses (zpng_lidarframe_encoder.cpp:52-73), with the sha256 of the plane
each must decode to.  Run in the build container (needs /root/reference for the build); the vectors
travel, the reference does not."""
import base64
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import zpng_ref  # noqa: E402

H, W = 32, 64   # the geometry the vectors' decoder must be created with


def planes():
    rng = np.random.default_rng(0x5eed)
    col = np.arange(W, dtype=np.uint64)[None, :]
    row = np.arange(H, dtype=np.uint64)[:, None]
    for dt, bits in ((np.uint8, 8), (np.uint16, 16), (np.uint32, 20), (np.uint64, 64)):
        hi = (1 << bits) - 1
        yield f"random_{np.dtype(dt).name}", rng.integers(0, hi, (H, W), dtype=np.uint64, endpoint=True).astype(dt)
        smooth = (row * 977 + col * 131 + (row * col) % 17) & np.uint64(hi)   # small left deltas, like a range image
        yield f"smooth_{np.dtype(dt).name}", smooth.astype(dt)
    yield "zeros_uint32", np.zeros((H, W), np.uint32)
    yield "ones_uint16", np.full((H, W), 0xFFFF, np.uint16)


def main():
    if not zpng_ref.available():
        sys.exit("oracle/_ref/libzpng_ref.so missing: run `make -C oracle` where /root/reference exists")
    out = {"h": H, "w": W, "source": "ZPNG_Compress of /root/reference/thirdparty/zpng/zpng.cpp", "vectors": {}}
    for name, p in planes():
        blob = zpng_ref.compress(p)
        px, w, h, ch, bpc = zpng_ref.decompress(blob)          # the reference's own inverse agrees
        assert (w, h) == (W, H) and px == p.tobytes(), name
        out["vectors"][name] = {"dtype": p.dtype.name, "zpng": base64.b64encode(blob).decode(),
                                "sha256": hashlib.sha256(p.tobytes()).hexdigest()}
    path = os.path.join(HERE, "osf", "zpng_ref_vectors.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, {k: len(v["zpng"]) for k, v in out["vectors"].items()})


if __name__ == "__main__":
    main()
