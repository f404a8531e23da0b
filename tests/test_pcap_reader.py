"""CPU-only: the C++ PcapReader (include/ouster/pcap/pcap.h) against the
reference's own captures and against hand-built pcaps with VLAN tags, Linux-cooked framing,
nanosecond timestamps and IPv4 fragmentation (ouster_pcap/src/ip_reassembler.cpp)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from conftest import PCAPS
from ouster_sdk_amd import _capi as capi


def _read(path, port=0, size=0, cap=8 << 20):
    L = capi.load_core()
    L.ouster_pcap_read_udp.argtypes = [C.c_char_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t,
                                       C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t,
                                       C.POINTER(C.c_int)]
    out = np.zeros(cap, dtype=np.uint8)
    sizes = np.zeros(4096, dtype=np.uint32)
    ports = np.zeros(4096, dtype=np.int32)
    msg = C.create_string_buffer(256)
    n = L.ouster_pcap_read_udp(path.encode(), port, size, out.ctypes.data, cap, sizes.ctypes.data,
                               ports.ctypes.data, 4096, msg, 256, C.byref(_read.truncated))
    if n < 0:
        raise RuntimeError(msg.value.decode())
    res, off = [], 0
    for i in range(n):
        res.append(out[off:off + sizes[i]].copy())
        off += int(sizes[i])
    return res, ports[:n]


_read.truncated = C.c_int(0)


def test_truncation_is_reported():
    """the reader used to stop silently when the caller's buffer was full."""
    path = os.path.join(PCAPS, "OS-2-128-U1_v2.3.0_1024x10.pcap")
    full, _ = _read(path)
    assert _read.truncated.value == 0 and len(full) > 4
    part, _ = _read(path, cap=sum(len(p) for p in full[:3]) + 10)
    assert len(part) == 3 and _read.truncated.value == 1


@pytest.mark.parametrize("base", ["OS-2-128-U1_v2.3.0_1024x10", "OS-0-32-U1_v2.2.0_1024x10",
                                  "OS-1-128_767798045_1024x10_20230712_120049", "crc_test"])
def test_matches_python_reader_on_reference_captures(oracle, base):
    O = oracle
    cal = O.calib_from_json(os.path.join(PCAPS, base + ".json"))
    pf = cal.packet_format()
    want = O.lidar_packets_from_pcap(os.path.join(PCAPS, base + ".pcap"), pf)
    got, _ = _read(os.path.join(PCAPS, base + ".pcap"), 7502, pf.lidar_packet_size)
    assert len(got) == len(want) > 0
    assert all(np.array_equal(g, w) for g, w in zip(got, want))
    allp, ports = _read(os.path.join(PCAPS, base + ".pcap"))
    assert set(ports.tolist()) <= {7502, 7503} and len(allp) >= len(got)


def _ipv4(payload, ident=1, flags_frag=0, proto=17):
    hdr = struct.pack(">BBHHHBBH4s4s", 0x45, 0, 20 + len(payload), ident, flags_frag, 64, proto, 0,
                      bytes([10, 0, 0, 1]), bytes([10, 0, 0, 2]))
    return hdr + payload


def _udp(data, sport=4000, dport=7502):
    return struct.pack(">HHHH", sport, dport, 8 + len(data), 0) + data


def _pcap(records, linktype=1, magic=0xA1B2C3D4):
    out = struct.pack("<IHHiIII", magic, 2, 4, 0, 0, 65535, linktype)
    for i, r in enumerate(records):
        out += struct.pack("<IIII", 100 + i, 5000 * i, len(r), len(r)) + r
    return out


def test_vlan_sll_nanos_and_fragment_reassembly(tmp_path):
    eth = bytes(12) + b"\x08\x00"
    vlan = bytes(12) + b"\x81\x00\x00\x05\x08\x00"
    a = bytes(range(200)) * 20          # 4000 B datagram, sent as three fragments
    whole = _udp(a)
    f1, f2, f3 = whole[:1480], whole[1480:2960], whole[2960:]
    recs = [
        eth + _ipv4(_udp(b"plain-udp")),
        vlan + _ipv4(_udp(b"behind-a-vlan-tag", dport=7503)),
        eth + _ipv4(f2, ident=7, flags_frag=(1480 // 8) | 0x2000),      # out of order
        eth + _ipv4(f1, ident=7, flags_frag=0x2000),
        eth + _ipv4(b"\x00" * 20, proto=6),                               # TCP: ignored
        eth + _ipv4(f3, ident=7, flags_frag=(2960 // 8)),
    ]
    p = tmp_path / "t.pcap"
    p.write_bytes(_pcap(recs))
    got, ports = _read(str(p))
    assert [bytes(g) for g in got] == [b"plain-udp", b"behind-a-vlan-tag", a]
    assert ports.tolist() == [7502, 7503, 7502]
    # linux cooked capture + nanosecond magic
    sll = bytes(14) + b"\x08\x00"
    p2 = tmp_path / "sll.pcap"
    p2.write_bytes(_pcap([sll + _ipv4(_udp(b"cooked"))], linktype=113, magic=0xA1B23C4D))
    got, _ = _read(str(p2))
    assert [bytes(g) for g in got] == [b"cooked"]
    with pytest.raises(RuntimeError, match="not a classic pcap"):
        bad = tmp_path / "bad.pcap"
        bad.write_bytes(b"\x0a\x0d\x0d\x0a" + bytes(40))
        _read(str(bad))


def test_incomplete_fragments_expire(tmp_path):
    """A lossy capture: thousands of datagrams that each lost their last fragment.  Their buffers are forgotten
    after a while (the reader's memory stays bounded), complete datagrams keep flowing, and an IP id that comes
    round again long after its first, incomplete use does not inherit the stale fragment."""
    eth = bytes(12) + b"\x08\x00"
    first = _udp(bytes(2000))[:1480]
    recs = []
    for i in range(6000):                                   # never completed: MF set, nothing follows
        recs.append(eth + _ipv4(first, ident=i % 65536, flags_frag=0x2000))
        if i % 1000 == 999:
            recs.append(eth + _ipv4(_udp(b"alive-%d" % i)))
    # IP id 5 again, 6000 packets later: a different datagram whose SECOND fragment arrives first
    b = bytes([7]) * 3000
    whole = _udp(b)
    recs.append(eth + _ipv4(whole[1480:], ident=5, flags_frag=(1480 // 8)))
    recs.append(eth + _ipv4(whole[:1480], ident=5, flags_frag=0x2000))
    p = tmp_path / "lossy.pcap"
    p.write_bytes(_pcap(recs))
    got, _ = _read(str(p))
    assert [bytes(g) for g in got] == [b"alive-%d" % i for i in range(999, 6000, 1000)] + [b]


# ---------------------------------------------------------------------------------------------------------------------
# IndexedPcapReader: which sensor sent a datagram, where every sensor's frames start (include/ouster/pcap/indexed_pcap_reader.h)
# ---------------------------------------------------------------------------------------------------------------------
def _infos(*names):
    from ouster_sdk_amd import core
    return [core.SensorInfo(open(os.path.join(PCAPS, n)).read()) for n in names]


def test_two_sensors_sharing_a_port_are_told_apart_by_serial_number(oracle):
    """same_ports_nonlegacy.pcap (the reference's fixture): two sensors send to port 7502, every lidar packet is a 16 640-byte
    datagram in twelve IP fragments.  Each is attributed to the sensor whose metadata carries the serial number of its packet
    header, and seek(frame offset) + next_packet() reads the reassembled packet again."""
    from ouster_sdk_amd import core
    O = oracle
    metas = ("same_ports_nonlegacy.1.json", "same_ports_nonlegacy.2.non_colliding_imu.json")
    infos = _infos(*metas)
    r = core.index_pcap(os.path.join(PCAPS, "same_ports_nonlegacy.pcap"), infos)
    lidar = [(idx, np.frombuffer(p, np.uint8), off) for idx, p, port, ts, off in r["packets"] if len(p) > 48]
    assert sorted(idx for idx, _, _ in lidar) == [0, 1] and all(port == 7502 for _, _, port, _, _ in r["packets"])
    for idx, pkt, off in lidar:
        cal = O.calib_from_json(os.path.join(PCAPS, metas[idx]))
        assert pkt.size == cal.packet_format().lidar_packet_size
        assert r["frame_offsets"][idx] == [off]                  # one frame per sensor, starting at this datagram
        assert core.PacketFormat(infos[idx]).prod_sn(pkt) == infos[idx].sn
        assert core.PacketFormat(infos[idx]).prod_sn(pkt) != infos[1 - idx].sn
        # the frame offset points at the FIRST fragment: reading from there yields the same payload
        again = core.read_pcap_udp(os.path.join(PCAPS, "same_ports_nonlegacy.pcap"), offset=off, max_packets=1)
        assert len(again) == 1 and np.array_equal(np.frombuffer(again[0][0], np.uint8), pkt)
    assert r["ports"] == [(7502, 7502), (7502, 7503)]


def test_indistinguishable_streams_on_one_port_are_refused():
    from ouster_sdk_amd import core
    for stem in ("same_ports", "same_ports_legacy", "same_ports_nonlegacy"):
        with pytest.raises(RuntimeError, match=r"Duplicate (lidar|imu) port/sn found in pcap: LEGACY_(IMU|LIDAR):750[23]"):
            core.index_pcap(os.path.join(PCAPS, stem + ".pcap"), _infos(stem + ".1.json", stem + ".2.json"))


def test_single_sensor_capture_index_and_guessed_ports():
    """One sensor: every lidar / IMU datagram belongs to it, the one frame starts at the first lidar packet, and ports the
    metadata does not name are guessed from the payload sizes seen in the capture."""
    from ouster_sdk_amd import core
    base = "OS-2-128-U1_v2.3.0_1024x10"
    info = _infos(base + ".json")[0]
    r = core.index_pcap(os.path.join(PCAPS, base + ".pcap"), [info])
    assert {idx for idx, *_ in r["packets"]} == {0}
    first_lidar = next(off for idx, p, port, ts, off in r["packets"] if port == 7502)
    assert r["frame_offsets"] == [[first_lidar]] and r["ports"] == [(7502, 7503)]
    # a packet of another sensor (other init_id / serial number) is not attributed ... unless ids are checked softly
    other = _infos("OS-0-128-U1_v2.3.0_1024x10.json")[0]
    other.format = info.format                          # same packet size, different ids
    r2 = core.index_pcap(os.path.join(PCAPS, base + ".pcap"), [other])
    assert all(idx is None for idx, p, port, *_ in r2["packets"] if port == 7502) and r2["frame_offsets"] == [[]]
    r3 = core.index_pcap(os.path.join(PCAPS, base + ".pcap"), [other], soft_id_check=True)
    assert all(idx == 0 for idx, p, port, *_ in r3["packets"] if port == 7502)


def test_reference_pcap_test_binary_passes():
    """The reference's own tests/pcap_test.cpp (PcapReader offsets / seek, IndexedPcapReader), compiled from where it lies
    by oracle/Makefile against include/ouster/pcap and linked with this repo's library; host only."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "oracle", "_ref", "cpptests", "pcap_test")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/cpptests/pcap_test not built (needs /root/reference at build time)")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, DATA_DIR=PCAPS))
    assert p.returncode == 0 and "[  PASSED  ] 11 tests." in p.stdout, p.stdout[-3000:] + p.stderr[-1000:]
