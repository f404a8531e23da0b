"""`ouster.sdk.core` for the reference's Python tests: ouster_sdk_amd.core re-exported under the reference's names,
plus the three things that are Python-side in the reference too: SensorInfo(json), stagger(), Packets."""
from ouster_sdk_amd.core import *  # noqa: F401,F403
from ouster_sdk_amd import core as _core
from ouster_sdk_amd.metadata import sensor_info_from_json as _from_json


class _SensorInfoMeta(type):
    """core.SensorInfo(json_text) builds one; isinstance(x, core.SensorInfo) keeps working."""
    def __call__(cls, *args, **kwargs):
        if len(args) == 1 and isinstance(args[0], str):
            return _from_json(args[0])
        return _core.SensorInfo(*args, **kwargs)

    def __instancecheck__(cls, obj):
        return isinstance(obj, _core.SensorInfo)


class SensorInfo(metaclass=_SensorInfoMeta):
    pass


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


def stagger(info, field):
    """python/src/ouster/sdk/core/data.py: stagger = destagger(..., inverse=True)."""
    return _core.destagger(info, field, inverse=True)


class Packets:
    """A list of packets with its metadata, iterated as (sensor index, packet) like core.Packets."""
    def __init__(self, packets, info):
        self._packets = list(packets)
        self.sensor_info = [info]

    def __iter__(self):
        return iter((0, p) for p in self._packets)
