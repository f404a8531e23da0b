"""`ouster.sdk`: the sub-packages of the reference's Python SDK this repo stands behind (core, pcap; util / sensor only as far
as the hot path's callers need the names)."""
from . import core  # noqa: F401
from . import pcap  # noqa: F401


def open_packet_source(*args, **kwargs):
    raise NotImplementedError("open_packet_source (source discovery / IO routing) is out of scope of this repo")
