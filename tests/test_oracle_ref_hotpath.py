"""The CPU baseline bench.py reports (cpu_baseline.kind == "reference") is the REFERENCE's own loops run by
oracle/hotpath_ref.cpp over a pool of frames (oracle/_ref/libhotpath_ref.so drives libdecode_ref.so and libcore_ref.so; it
holds no reference code).  Checked here: what that harness computes is what the oracle computes from the same packets --
every plane, and the cloud -- on one thread and with the frames spread over threads; and the OpenMP build of the reference's
cartesianT (-DOUSTER_OMP, the reference's own parallel form) gives the same cloud as the serial one."""
import ctypes as C

import numpy as np
import pytest

from oracle import hotpath_ref

pytestmark = pytest.mark.skipif(not hotpath_ref.available(), reason="oracle/_ref is built where /root/reference exists")


def _setup(O, profile="RNG15_RFL8_NIR8_DUAL", h=32, w=256, frames=3):
    cal = O.synthetic_calib(h=h, w=w, profile=profile)
    pf = cal.packet_format()
    packets, _ = O.synth_packets(cal, frames, seed=0xC0FFEE)
    pool = np.ascontiguousarray(np.stack([np.stack(list(p)) if not isinstance(p, np.ndarray) else p for p in packets]))
    ldir, lofs = cal.xyz_lut(False)
    shifts = np.array([(3, 1, -1, -3)[i % 4] for i in range(h)], dtype=np.int32)
    fr = O.Frame.for_profile(cal.profile, h, w, cal.cpp, with_window=True)
    dtypes = {n: fr.plane(n).dtype for n in fr.plane_names()}
    return cal, pf, pool, ldir, lofs, shifts, fr, dtypes


@pytest.mark.parametrize("threads", [1, 2])
def test_reference_harness_equals_oracle(oracle, threads):
    O = oracle
    cal, pf, pool, ldir, lofs, shifts, fr, dtypes = _setup(O)
    hp = hotpath_ref.HotPath(O, pf, pool, dtypes, ["RANGE", "RANGE2", "REFLECTIVITY", "REFLECTIVITY2"], ["RANGE", "RANGE2"],
                             ldir, lofs, shifts, block_dim=16)
    n = pool.shape[0]
    # one pass over the pool: with a static schedule thread 0's last frame is the last frame of its contiguous share
    t, legs, planes, cloud = hp.run(n, 1, threads=threads, own_inputs=threads > 1, want_outputs=True)
    assert t > 0 and all(x >= 0 for x in legs)
    last = (n + threads - 1) // threads - 1
    O.batch_frame(pf, pool[last], fr, init_id=O.lib().ora_init_id(C.byref(pf), pool[last][0].ctypes.data))
    for name, got in planes.items():
        assert np.array_equal(got, fr.plane(name)), name
    assert np.array_equal(cloud, O.cartesian(fr.plane("RANGE"), ldir, lofs))


def test_reference_omp_cartesian_equals_serial(oracle):
    if not hotpath_ref.omp_available():
        pytest.skip("libcore_ref_omp.so not built")
    O = oracle
    from oracle import core_ref
    cal, pf, pool, ldir, lofs, shifts, fr, dtypes = _setup(O)
    O.batch_frame(pf, pool[0], fr, init_id=O.lib().ora_init_id(C.byref(pf), pool[0][0].ctypes.data))
    serial = core_ref.cartesian(fr.plane("RANGE"), ldir, lofs)
    omp = C.CDLL(hotpath_ref.CORE_OMP)
    omp.ref_cartesian_f64.restype = None
    omp.ref_cartesian_f64.argtypes = [C.c_void_p] * 4 + [C.c_size_t] * 2
    r = np.ascontiguousarray(fr.plane("RANGE"), dtype=np.uint32)
    pts = np.full((r.size, 3), np.nan)
    omp.ref_cartesian_f64(pts.ctypes.data, r.ctypes.data, ldir.ctypes.data, lofs.ctypes.data, r.shape[0], r.shape[1])
    assert np.array_equal(pts, serial)
    assert hotpath_ref.bench_cartesian_omp(r, ldir, lofs, 2, 2) > 0
