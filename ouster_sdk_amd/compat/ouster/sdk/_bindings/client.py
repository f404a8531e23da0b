"""`ouster.sdk._bindings.client`: the low-level names the reference's tests import from its compiled module."""
from ouster_sdk_amd.core import (frame_to_packets, SensorInfo, LidarFrame, PacketFormat, FrameBatcher, FieldType,  # noqa: F401
                                 get_field_types, LidarPacket)


class FrameSet(list):
    """The frames of one tick, one per sensor (a collation of LidarFrames in the reference)."""


class FrameSetSource:
    """Base of the reference's frame sources; tests/multi.py derives its packet-batching `Frames` from it."""
    def __init__(self):
        pass

    def close(self):
        self._source = None
