"""Property tests (hypothesis) of the oracle's restatements, CPU only.

* field_info + FieldDecodeInfo::get/set (parsing.cpp:57-122, field_decode_info.h:41-78): for any bit
  field that fits, `get` returns exactly the bits written into the buffer (little-endian bit order,
  upshift applied), `set` is its inverse and touches no bit outside the field, and the end-of-buffer
  clamp keeps every 8-byte window inside `max_length`.
* destagger (impl/lidar_frame_impl.h:733-760): equals the reference's own arithmetic
  `(w + size_t(sign*shift) % w) % w` -- which is np.roll only while the size_t wrap is harmless -- and
  inverse undoes it exactly in that regime.
"""
import ctypes as C

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st


@st.composite
def bit_fields(draw):
    bit_size = draw(st.integers(1, 56))
    upshift = draw(st.integers(0, 64 - bit_size - 7 if bit_size < 57 else 0))
    max_length = draw(st.integers(9, 40))
    # the field (and the 8-byte read window after the offset clamp) must fit the buffer
    bit_start = draw(st.integers(0, max_length * 8 - bit_size))
    return bit_start, bit_size, upshift, max_length


@settings(max_examples=300, deadline=None)
@given(bit_fields(), st.integers(0, 2 ** 64 - 1), st.binary(min_size=40, max_size=40))
def test_field_info_get_set(oracle, bf, value, noise):
    O = oracle
    bit_start, bit_size, upshift, max_length = bf
    needs = bit_size + upshift
    size_bytes = (needs + 7) // 8
    if bit_start // 8 + size_bytes > max_length:
        with pytest.raises(ValueError):
            O.field_info(bit_start, bit_size, upshift, max_length)
        return
    f = O.field_info(bit_start, bit_size, upshift, max_length)
    assert 0 <= f.offset and f.offset + 8 <= max_length   # the 8-byte window never leaves the buffer
    if bit_start // 8 + 8 > max_length:
        assert f.offset + 8 == max_length           # clamped against the end
    # the mask, moved back to absolute bit positions, covers exactly the field
    absolute = (int(f.mask) << (8 * f.offset)) & ((1 << (8 * (f.offset + 8))) - 1)
    assert absolute == ((1 << bit_size) - 1) << bit_start
    buf = np.frombuffer(noise, dtype=np.uint8).copy()
    raw = int.from_bytes(buf.tobytes(), "little")
    want = ((raw >> bit_start) & ((1 << bit_size) - 1)) << upshift
    got = O.lib().ora_fdi_get(C.byref(f), buf.ctypes.data)
    assert got == want & (2 ** 64 - 1)
    # set is the inverse of get and leaves every other bit alone
    v = (value & ((1 << bit_size) - 1)) << upshift
    O.lib().ora_fdi_set(C.byref(f), buf.ctypes.data, v)
    assert O.lib().ora_fdi_get(C.byref(f), buf.ctypes.data) == v
    after = int.from_bytes(buf.tobytes(), "little")
    field_bits = ((1 << bit_size) - 1) << bit_start
    assert (after & ~field_bits) == (raw & ~field_bits)


@settings(max_examples=150, deadline=None)
@given(st.integers(1, 12), st.integers(1, 40), st.data())
def test_destagger_offsets(oracle, h, w, data):
    O = oracle
    shifts = data.draw(st.lists(st.integers(-3 * w, 3 * w), min_size=h, max_size=h))
    inverse = data.draw(st.booleans())
    img = np.arange(h * w, dtype=np.uint16).reshape(h, w)
    got = O.destagger(img, shifts, inverse)
    sign = -1 if inverse else 1
    for u in range(h):
        x = (sign * shifts[u]) % (2 ** 64)              # size_t(sign * shift)
        offset = (w + x % w) % w
        want = np.concatenate([img[u, w - offset:], img[u, :w - offset]]) if offset else img[u]
        assert np.array_equal(got[u], want), (u, shifts[u], offset)
        if (2 ** 64) % w == 0 or sign * shifts[u] >= 0:  # no harm from the unsigned wrap: it is np.roll
            assert np.array_equal(got[u], np.roll(img[u], sign * shifts[u]))
    if all(((2 ** 64) % w == 0) or s == 0 for s in shifts) or (2 ** 64) % w == 0:
        assert np.array_equal(O.destagger(got, shifts, not inverse), img)
