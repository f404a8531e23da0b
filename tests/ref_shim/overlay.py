"""Test infrastructure for tests/test_reference_python_tests.py: the two modules of `ouster.sdk` that the reference's tests
import and that ARE the reference's own code, registered from where oracle/Makefile staged them (oracle/_ref/pytests,
git-ignored) -- they are the independent side of those tests' comparisons, so they must not be this repo's:
  ouster.sdk.examples.reference   python/src/ouster/sdk/examples/reference.py (the closed-form xyz LUT the tests compare with)
  ouster.sdk.core._digest         python/src/ouster/sdk/core/_digest.py (md5 digests of fields / planes vs the *_digest.json)
Everything else of `ouster.sdk` is the product's ouster_sdk_amd/compat/ouster package."""
import importlib.util
import os
import sys
import types

STAGED = os.environ.get("OUSTER_REF_STAGED", "")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def install():
    import ouster.sdk as sdk
    import ouster.sdk.core as core
    ex = types.ModuleType("ouster.sdk.examples")
    ex.__path__ = [os.path.join(STAGED, "examples")]
    sys.modules["ouster.sdk.examples"] = ex
    sdk.examples = ex
    ref = os.path.join(STAGED, "examples", "reference.py")
    if os.path.exists(ref):
        ex.reference = _load("ouster.sdk.examples.reference", ref)
    dig = os.path.join(STAGED, "core", "_digest.py")
    if os.path.exists(dig):
        core._digest = _load("ouster.sdk.core._digest", dig)
