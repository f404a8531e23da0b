"""The REFERENCE's own Python tests of this path, run UNMODIFIED against this repo:
/root/reference/python/tests/test_xyzlut.py, test_destagger.py, test_batching.py, test_parsing.py, test_data.py, test_core.py and test_extended_profiles.py (with the reference helpers they
import: tests/multi.py and ouster/sdk/core/_digest.py) -- staged verbatim by oracle/Makefile into the
git-ignored oracle/_ref/pytests where the reference checkout exists (it travels to the GPU box with the snapshot) -- are
collected by a child pytest whose `ouster.sdk.core` is the product's ouster_sdk_amd/compat/ouster (= ouster_sdk_amd.core + the JSON metadata reader
ouster_sdk_amd/metadata.py) and whose fixtures (tests/ref_shim/conftest_for_reference_tests.py) mirror the reference's conftest.  Every
collected test must pass; the counts are printed."""
import os
import re
import shutil
import subprocess
import sys

import pytest

from conftest import PCAPS

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGED = os.path.join(ROOT, "oracle", "_ref", "pytests")
SHIM = os.path.join(ROOT, "tests", "ref_shim")            # conftest, overlay, more_itertools stand-in
COMPAT = os.path.join(ROOT, "ouster_sdk_amd", "compat")   # the product's `ouster.sdk` package


# test id fragments that are left out, and why (everything else that is collected must pass)
OUT_OF_SCOPE = {
    "test_batching_dups": "IMU packets batched into the frame (ImuPacket, ACCEL32_GYRO32_NMEA): out of scope, SURVEY section 8",
    "test_packet_overheat": "open_packet_source (source discovery / IO routing): out of scope",
    "test_make_packets": "ImuPacket / ZonePacket: out of scope",
    "test_imu_packet": "IMU packet accessors: out of scope",
    "test_lidar_frame_zones_access": "zone-monitoring states carried by a frame: out of scope",
    "test_sensor_": "live sensor sockets (ouster.sdk.sensor): out of scope",
    "test_frames_closed": "needs a live SensorPacketSource: out of scope",
    "test_version_": "core.Version parsing: out of scope",
    "test_pointcloud_load": "point cloud file IO: out of scope",
    "test_voxel_downsample": "voxel down-sampling: out of scope",
}
FILES = ("test_xyzlut.py", "test_destagger.py", "test_batching.py", "test_parsing.py", "test_data.py", "test_core.py",
         "test_extended_profiles.py")
HELPERS = ("multi.py",)     # tests/multi.py of the reference: its packet-batching `Frames` source, used by test_core.py
# files of which only the named tests are in scope (the rest of test_pcap.py records captures, reads IMU packets, indexes
# and seeks through the reference's Python pcap package): real captures read back, ports inferred from the capture
ONLY = {"test_pcap.py": ("test_pcap_read_real", "test_pcap_guess_real")}


@pytest.mark.skipif(not os.path.exists(os.path.join(STAGED, "test_batching.py")),
                    reason="oracle/_ref/pytests is staged by `make -C oracle` only where /root/reference exists")
def test_reference_python_tests_pass_unmodified(tmp_path):
    """test_xyzlut.py, test_destagger.py, test_batching.py, test_parsing.py, test_data.py, test_core.py and test_extended_profiles.py of the reference, byte-identical, in a
    package laid out like the reference's (`tests/conftest.py`, `from tests.conftest import PCAPS_DATA_DIR`)."""
    pkg = tmp_path / "tests"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    for name in FILES + HELPERS + tuple(ONLY):
        shutil.copy(os.path.join(STAGED, name), pkg / name)          # byte-identical copies
    shutil.copy(os.path.join(SHIM, "conftest_for_reference_tests.py"), pkg / "conftest.py")
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([COMPAT, SHIM, str(tmp_path), ROOT, env.get("PYTHONPATH", "")])
    env["OUSTER_REF_PCAPS"] = PCAPS
    env["OUSTER_REF_STAGED"] = STAGED          # the shim's ouster.sdk.core._digest executes the staged reference module
    deselect = " and ".join("not " + k for k in OUT_OF_SCOPE)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "--rootdir", str(tmp_path),
                        "-c", os.devnull, "-k", deselect] + [str(pkg / n) for n in FILES] +
                       [f"{pkg / n}::{t}" for n, ts in ONLY.items() for t in ts],
                       env=env, capture_output=True, text=True, timeout=1500)
    tail = "\n".join(r.stdout.strip().splitlines()[-40:])
    m = re.search(r"(\d+) passed", r.stdout)
    passed = int(m.group(1)) if m else 0
    print(f"reference python tests: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]}")
    assert r.returncode == 0 and passed >= 148 and "failed" not in r.stdout.splitlines()[-1], tail + r.stderr[-2000:]
