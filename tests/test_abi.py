"""CPU-only checks of the drop-in boundary: the C ABI library loads, exports every symbol
include/ouster_hip.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, has_gpu
from ouster_sdk_amd import _capi as capi


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ouster_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ouster_hip_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported():
    names = _declared_symbols()
    assert len(names) >= 17
    L = capi.load_hip()
    for n in names:
        assert hasattr(L, n), n
    assert sorted(capi.ABI_SYMBOLS) == names
    assert b"gfx950" in L.ouster_hip_version()


def test_struct_layouts_match_header():
    # sizes the C compiler produces for the PODs (LP64): guards the ctypes mirrors
    assert C.sizeof(capi.Bits) == 16
    assert C.sizeof(capi.FieldDesc) == 24
    assert C.sizeof(capi.FormatDesc) == 10 * 4 + 9 * 16 + 8 + 32 * 24
    assert C.sizeof(capi.FrameMeta) == 24
    assert C.sizeof(capi.OsfPlane) == 32
    assert C.sizeof(capi.FrameOut) == 2 * 32 * 8 + 6 * 8 + 2 * 8 + 4 * 4 + 8 + 4 * 4 + 8   # + the range-gate by-product + xyz_poses
    assert C.sizeof(capi.Calib) == 8 + 8 + 128 + 128 + 8 + 8 + 8


def test_host_library_loads_and_builds_descs():
    L = capi.load_core()
    assert hasattr(L, "ouster_core_format_desc")
    planes = capi.default_planes("RNG15_RFL8_NIR8_DUAL", True)
    assert [p[0] for p in planes] == ["RANGE", "REFLECTIVITY", "NEAR_IR", "RANGE2", "REFLECTIVITY2",
                                     "FLAGS", "FLAGS2", "WINDOW"]
    assert sum(p[1] for p in planes) == 15  # bytes per pixel of the metric's plane set
    d = capi.format_desc("RNG15_RFL8_NIR8_DUAL", 128, 16, 2048, planes)
    assert (d.lidar_packet_size, d.col_size, d.channel_data_size) == (16640, 1036, 8)
    with pytest.raises(ValueError, match="Unknown lidar udp profile"):
        capi.format_desc("NOT_A_PROFILE", 128, 16, 2048, planes)
    with pytest.raises(ValueError, match="Dest type too small"):
        capi.format_desc("RNG15_RFL8_NIR8_DUAL", 128, 16, 2048, [("RANGE", 2)])


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly():
    with pytest.raises(capi.OusterHipError):
        capi.Context(0)
    msg = capi.load_hip().ouster_hip_last_error().decode()
    assert "HIP device" in msg or "device" in msg
    from ouster_sdk_amd import device
    with pytest.raises(capi.OusterHipError, match="no CPU fallback"):
        device.HotPath("RNG15_RFL8_NIR8_DUAL", 128, 2048)


def test_host_pool_entry_points_work_without_a_gpu_as_plain_memory():
    """include/ouster_hip.h, "host containers": without a GPU ouster_hip_host_alloc hands out plain heap memory (the C++
    containers still work), nothing is reported as GPU-addressable, the counters stay at zero and the compute entry points on
    host arrays refuse like every other one (no CPU fallback).  With a GPU the same calls are exercised by the C++ drop-in test."""
    L = capi.load_hip()
    st0 = capi.alloc_stats()
    for n in (1, 100, 4096, 1 << 20):
        p = L.ouster_hip_host_alloc(n, 1)
        assert p
        buf = (C.c_uint8 * n).from_address(p)
        assert not any(buf[:: max(1, n // 64)])          # zeroed
        buf[n - 1] = 7
        if not has_gpu():
            assert L.ouster_hip_host_is_pinned(p, n) == 0
        elif n >= 2048:
            assert L.ouster_hip_host_is_pinned(p, n) == 1 and L.ouster_hip_host_is_pinned(p + 1, n) == 0
        L.ouster_hip_host_free(p)
    L.ouster_hip_host_free(None)
    L.ouster_hip_host_pool_trim(0)
    st1 = capi.alloc_stats()
    if not has_gpu():
        assert st1["pinned_allocs"] == st0["pinned_allocs"] == 0 and st1["device_allocs"] == 0 and st1["pool_requests"] == 0
        a = (C.c_uint32 * 16)()
        assert L.ouster_hip_destagger_host(None, a, a, 4, 4, 4, a, 4, 0) == capi.ERR_INVALID_ARGUMENT   # no context can exist
