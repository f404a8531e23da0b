"""`ouster.sdk._bindings.client`: the low-level names the reference's tests import from its compiled module."""
from ouster_sdk_amd.core import frame_to_packets  # noqa: F401
