"""`ouster.sdk` for the reference's Python tests: the sub-packages this repo can stand behind."""
from . import core  # noqa: F401
from . import pcap  # noqa: F401


def open_packet_source(*args, **kwargs):
    raise NotImplementedError("open_packet_source (source discovery / IO routing) is out of scope of this repo")
