"""`ouster.sdk.pcap` for the reference's Python tests, on this repo's capture reader (ouster_sdk_amd.core.read_pcap_udp,
include/ouster/pcap/pcap.h): lidar packets of a capture with their metadata, and frames assembled from them by the
GPU-backed FrameBatcher.  IMU / zone packets are passed over (out of scope); indexing, seeking and multi-sensor routing
of the reference's pcap package are not mirrored."""
from ouster_sdk_amd import core as _core
from ouster.sdk.util import resolve_metadata


class PacketInfo:
    def __init__(self):
        self.dst_ip = self.src_ip = ""
        self.dst_port = self.src_port = 0
        self.payload_size = self.packet_size = 0
        self.timestamp = 0.0
        self.fragments_in_packet = 1
        self.ip_version = 4
        self.encapsulation_protocol = 0
        self.network_protocol = 17


def _infos(path, sensor_info):
    if sensor_info:
        return list(sensor_info)
    meta = resolve_metadata(path)
    if meta is None:
        raise RuntimeError("no metadata found next to " + path)
    with open(meta) as f:
        return [_core.SensorInfo(f.read())]


class PcapPacketSource:
    """Iterates (sensor index, packet); only lidar packets of the first sensor are produced."""

    def __init__(self, path, sensor_info=None, **_):
        import copy
        self._path = path
        self.sensor_info = [copy.copy(i) for i in _infos(path, sensor_info)]
        self._pf = _core.PacketFormat(self.sensor_info[0])
        # ports the metadata does not name are inferred from the capture (IndexedPcapReader: payload sizes per stream)
        try:
            ports = _core.index_pcap(path, self.sensor_info)["ports"]
        except RuntimeError:
            ports = []
        for info, (lidar, imu) in zip(self.sensor_info, ports):
            if info.config.udp_port_lidar is None:
                info.config.udp_port_lidar = lidar
            if info.config.udp_port_imu is None:
                info.config.udp_port_imu = imu

    def __iter__(self):
        fmt = self._pf
        for payload, port, ts in _core.read_pcap_udp(self._path):
            if len(payload) != fmt.lidar_packet_size:
                continue
            p = _core.LidarPacket(fmt)
            p.buf = payload
            p.host_timestamp = ts
            yield 0, p

    def close(self):
        pass


class PcapFrameSetSource:
    """Iterates frame sets (lists with one frame per sensor); the frame still open at the end of the capture is
    delivered too, as the reference's source does."""

    def __init__(self, path, sensor_info=None, **_):
        self._packets = PcapPacketSource(path, sensor_info=sensor_info)
        self.sensor_info = self._packets.sensor_info

    def __iter__(self):
        info = self.sensor_info[0]
        batcher, frame, open_frame = _core.FrameBatcher(info), _core.LidarFrame(info), False
        for _, p in self._packets:
            open_frame = True
            if batcher.batch(p, frame):
                yield [frame]
                frame, open_frame = _core.LidarFrame(info), False
        if open_frame:
            yield [frame]

    def close(self):
        pass
