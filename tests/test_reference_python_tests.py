"""The REFERENCE's own Python tests of this path, run UNMODIFIED against this repo:
/root/reference/python/tests/test_xyzlut.py and test_destagger.py -- staged verbatim by oracle/Makefile into the
git-ignored oracle/_ref/pytests where the reference checkout exists (it travels to the GPU box with the snapshot) -- are
collected by a child pytest whose `ouster.sdk.core` is tests/ref_shim (= ouster_sdk_amd.core + the JSON metadata reader
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
SHIM = os.path.join(ROOT, "tests", "ref_shim")


@pytest.mark.skipif(not os.path.exists(os.path.join(STAGED, "test_xyzlut.py")),
                    reason="oracle/_ref/pytests is staged by `make -C oracle` only where /root/reference exists")
def test_reference_xyzlut_and_destagger_tests_pass_unmodified(tmp_path):
    for name in ("test_xyzlut.py", "test_destagger.py"):
        shutil.copy(os.path.join(STAGED, name), tmp_path / name)          # byte-identical copies
    shutil.copy(os.path.join(SHIM, "conftest_for_reference_tests.py"), tmp_path / "conftest.py")
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([SHIM, ROOT, env.get("PYTHONPATH", "")])
    env["OUSTER_REF_PCAPS"] = PCAPS
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "--rootdir", str(tmp_path),
                        "-c", os.devnull, str(tmp_path)], env=env, capture_output=True, text=True, timeout=1500)
    tail = "\n".join(r.stdout.strip().splitlines()[-25:])
    m = re.search(r"(\d+) passed", r.stdout)
    passed = int(m.group(1)) if m else 0
    print(f"reference python tests: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]}")
    assert r.returncode == 0 and passed >= 27 and "failed" not in r.stdout.splitlines()[-1], tail + r.stderr[-2000:]
