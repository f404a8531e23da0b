"""Test shim: lets the REFERENCE's own Python tests (python/tests/test_xyzlut.py, test_destagger.py, staged by
oracle/Makefile into oracle/_ref/pytests where /root/reference exists) import `ouster.sdk.core` and get this repo's
implementation (ouster_sdk_amd.core).  See tests/test_reference_python_tests.py."""
