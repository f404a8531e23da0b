"""`ouster` -- the reference's top-level Python package name, served by this repo: put ouster_sdk_amd/compat on PYTHONPATH
and `from ouster.sdk import core` resolves to ouster_sdk_amd.core under the reference's names.
The reference's own Python tests run against it unmodified (tests/test_reference_python_tests.py)."""
