"""The oracle's cartesian and destagger restatements pinned on the REFERENCE'S OWN loops: impl/cartesian.h
and destagger_into<T> of impl/lidar_frame_impl.h compiled from /root/reference into oracle/_ref/libcore_ref.so
(oracle/Makefile, tests/core_ref.py).  Bit for bit: these are integer copies and one multiply-add per coordinate in the
LUT's own precision (both sides built with -ffp-contract=off, like a default build of the reference)."""
import numpy as np
import pytest

import core_ref

pytestmark = pytest.mark.skipif(not core_ref.available(),
                                reason="oracle/_ref/libcore_ref.so is built only where /root/reference exists "
                                       "(tests/test_evidence_manifest.py fails when it should be there and is not)")


@pytest.mark.parametrize("T", [np.float64, np.float32])
@pytest.mark.parametrize("h,w", [(128, 2048), (64, 1024), (16, 512), (32, 1000)])
def test_cartesian_equals_the_compiled_reference(oracle, T, h, w):
    O = oracle
    rng = np.random.default_rng(h * w)
    cal = O.synthetic_calib(h=h, w=w, profile="RNG19_RFL8_SIG16_NIR16")
    d64, o64 = cal.xyz_lut(True)
    d, o = d64.astype(T), o64.astype(T)
    r = rng.integers(0, 1 << 19, size=(h, w), dtype=np.uint32)
    r[rng.random((h, w)) < 0.3] = 0
    r[0, :7] = [0, 1, 2, (1 << 19) - 1, 0xFFFFFFFF, 1 << 31, 12345]     # extremes of the u32 range image
    want = core_ref.cartesian(r, d, o)
    got = O.cartesian(r, d, o)
    assert got.dtype == want.dtype == T and got.shape == want.shape == (h * w, 3)
    assert np.array_equal(got, want)                      # every bit
    assert not np.isnan(want).any() and np.all(want[r.reshape(-1) == 0] == 0)
    # and the LUT the oracle builds is what the reference's own example computes (tests/test_reference_python_tests.py runs
    # the reference's test_xyzlut.py); here: float LUT = the double LUT rounded once
    assert np.array_equal(d, d64.astype(T))


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.uint32, np.uint64])
@pytest.mark.parametrize("h,w", [(128, 2048), (16, 512), (32, 1000), (8, 999)])
def test_destagger_equals_the_compiled_reference(oracle, dtype, h, w):
    """Power-of-two widths and the others: the reference evaluates (w + sign * shift % w) % w with w as size_t, which is
    np.roll only for power-of-two widths -- the oracle must follow the reference, not numpy."""
    O = oracle
    rng = np.random.default_rng(h + w)
    img = rng.integers(0, np.iinfo(dtype).max, size=(h, w), dtype=dtype, endpoint=True)
    for shifts in (rng.integers(-3 * w, 3 * w, h), rng.integers(0, 40, h), -rng.integers(0, 40, h), np.zeros(h, int)):
        for inverse in (False, True):
            want = core_ref.destagger(img, shifts, inverse)
            got = O.destagger(img, shifts, inverse)
            assert np.array_equal(got, want), (shifts[:4], inverse)
    if w & (w - 1) == 0:   # power of two: a plain roll, and destagger o stagger = identity
        sh = rng.integers(-50, 50, h)
        assert np.array_equal(core_ref.destagger(img, sh), np.stack([np.roll(img[r], sh[r]) for r in range(h)]))
        assert np.array_equal(core_ref.destagger(core_ref.destagger(img, sh), sh, True), img)
    with pytest.raises(ValueError, match="image height does not match shifts size"):
        core_ref.destagger(img, np.zeros(h + 1, int))
    with pytest.raises(ValueError, match="image height does not match shifts size"):
        O.destagger(img, np.zeros(h + 1, int))
