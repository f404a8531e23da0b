"""Runs the C++ test executable of the host API mirror (tests/cpp/test_core_api.cpp) on the
GPU box: PacketFormat / LidarFrame / FrameBatcher / destagger / XYZLut with the reference's
names, every per-pixel result produced by the HIP kernels."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_cpp_core_api():
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "test_core_api")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", ROOT, "cpptests"])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "ouster_sdk_amd", "lib") + ":/opt/rocm/lib:" + \
        env.get("LD_LIBRARY_PATH", "")
    p = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    print(p.stdout[-4000:])
    print(p.stderr[-2000:])
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]
    assert "0 failed" in p.stdout
