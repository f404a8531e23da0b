"""`ouster.sdk.core`: ouster_sdk_amd.core re-exported under the reference's names (python/src/ouster/sdk/core/__init__.py),
plus the three things that are Python-side in the reference too: SensorInfo(json), stagger(), Packets."""
from ouster_sdk_amd.core import *  # noqa: F401,F403
from ouster_sdk_amd import core as _core


from .data import ColHeader  # noqa: E402,F401  (a Python Enum in the reference too)

Packet = _core.LidarPacket      # the base class of the reference's packet types; only lidar packets exist here
SensorInfo = _core.SensorInfo   # SensorInfo(json_text) is a constructor of the C++ class (csrc/host/metadata.cpp)


class ChanField:
    """Field-name constants (ouster_core/include/ouster/core/chanfield.h)."""
    RANGE = "RANGE"
    RANGE2 = "RANGE2"
    SIGNAL = "SIGNAL"
    SIGNAL2 = "SIGNAL2"
    REFLECTIVITY = "REFLECTIVITY"
    REFLECTIVITY2 = "REFLECTIVITY2"
    NEAR_IR = "NEAR_IR"
    FLAGS = "FLAGS"
    FLAGS2 = "FLAGS2"
    WINDOW = "WINDOW"
    RAW_HEADERS = "RAW_HEADERS"
    RAW32_WORD1 = "RAW32_WORD1"
    RAW32_WORD2 = "RAW32_WORD2"
    RAW32_WORD3 = "RAW32_WORD3"
    RAW32_WORD4 = "RAW32_WORD4"


def stagger(info, field):
    """python/src/ouster/sdk/core/data.py: stagger = destagger(..., inverse=True)."""
    return _core.destagger(info, field, inverse=True)


class Packets:
    """Packets with their metadata, iterated as (sensor index, packet) like core.Packets; the iterable is consumed lazily
    (the reference's tests hand over endless generators)."""
    def __init__(self, packets, info):
        self._packets = packets
        self.sensor_info = [info]

    def __iter__(self):
        return ((0, p) for p in self._packets)

    is_live = False
    is_indexed = False

    def close(self):
        pass


PacketSource = Packets


class ImuPacket:
    """IMU packets are out of scope of this repo; the name exists so that test modules import."""
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("IMU packets are out of scope")


class Version:
    """Firmware version parsing is not part of this repo's Python surface; the name exists so that test modules import."""
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("core.Version is out of scope")
