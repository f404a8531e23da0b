"""The three replay calls of the reference's low-level pcap binding that its parsing tests use, on this repo's reader."""
from ouster_sdk_amd import core as _core


class _Replay:
    def __init__(self, path):
        self.packets = list(_core.read_pcap_udp(path))
        self.pos = -1


def replay_initialize(path):
    return _Replay(path)


def next_packet_info(handle, info):
    handle.pos += 1
    if handle.pos >= len(handle.packets):
        return False
    payload, port, ts = handle.packets[handle.pos]
    info.dst_port = port
    info.payload_size = len(payload)
    info.timestamp = ts
    return True


def read_packet(handle, buf):
    payload = handle.packets[handle.pos][0]
    buf[:len(payload)] = payload
    return len(payload)
