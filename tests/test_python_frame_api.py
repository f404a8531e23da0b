"""Host-side pieces of the Python LidarFrame / FieldType / PacketFormat surface (no GPU): the forms the reference's
python/tests/test_data.py exercises (python/src/cpp/client/lidar_frame.cpp:273-370 add_field, :617-638 packet timestamps,
field.cpp:57-153 FieldType, packet.cpp:213-256 packet_header, :313-336 column setters)."""
import numpy as np
import pytest

from ouster_sdk_amd import core


def _info(h=32, w=1024, profile=core.UDPProfileLidar.RNG19_RFL8_SIG16_NIR16):
    info = core.SensorInfo()
    info.format.columns_per_packet = 16
    info.format.pixels_per_column = h
    info.format.columns_per_frame = w
    info.format.udp_profile_lidar = profile
    info.fw_rev = "3.2.1"
    return info


def test_add_field_forms():
    fr = core.LidarFrame(8, 64, [], 16)
    a = fr.add_field("by_dtype", np.int16, (3,), core.FieldClass.PIXEL_FIELD)
    assert a.shape == (8, 64, 3) and a.dtype == np.int16 and not a.any()
    b = fr.add_field("by_value", np.arange(8 * 64, dtype=np.float32).reshape(8, 64))
    assert b.dtype == np.float32 and b[1, 2] == 66.0
    c = fr.add_field(core.FieldType("by_type", np.uint64, (), core.FieldClass.COLUMN_FIELD))
    assert c.shape == (64,) and fr.field_class("by_type") == core.FieldClass.COLUMN_FIELD
    d = fr.add_field("frame_vec", np.zeros((5,), np.uint8), core.FieldClass.FRAME_FIELD)
    assert d.shape == (5,)
    assert fr.add_field("empty", np.int8, (0,), core.FieldClass.FRAME_FIELD).shape == (0,)
    with pytest.raises(ValueError, match="Duplicated field"):
        fr.add_field("by_dtype", np.int16)                               # duplicate
    assert fr.add_field("two_args", np.uint8).shape == (8, 64) and fr.add_field("str_dtype", "u2").dtype == np.uint16
    with pytest.raises(ValueError):
        fr.add_field("zero_px", np.int8, (0,))                           # a pixel field cannot be empty
    with pytest.raises(ValueError):
        fr.add_field("bad_shape", np.zeros((8, 63), np.uint8))           # pixel field must be h x w
    with pytest.raises(IndexError):
        fr.field("missing")
    assert fr.packet_count == 4


def test_copy_construction_casts_extends_and_retracts():
    src = core.LidarFrame(4, 32, [core.FieldType("a", np.uint32), core.FieldType("b", np.uint8)], 16)
    src.field("a")[:] = 2 ** 16 - 1
    src.field("b")[:] = 7
    dst = core.LidarFrame(src, [core.FieldType("a", np.uint8), core.FieldType("c", np.uint16)])
    assert dst.fields == ["a", "c"]
    assert dst.field("a").dtype == np.uint8 and (dst.field("a") == 255).all()
    assert not dst.field("c").any()
    same = core.LidarFrame(src)
    assert same == src and same is not src


def test_field_type_dtype_and_dims():
    ft = core.FieldType("X", np.uint32, (1, 2, 3))
    assert ft.extra_dims == (1, 2, 3) and ft.element_type is np.dtype(np.uint32)
    ft.extra_dims = (4,)
    ft.element_type = np.dtype("S30")
    assert ft.element_type == np.dtype("S1") and ft.extra_dims == (4, 30)
    ft.element_type = np.uint8
    assert ft.extra_dims == (4,)
    assert core.FieldType("X", np.uint8) != "X"


def test_min_max_valid_packet_timestamp_counts_other_streams():
    fr = core.LidarFrame(_info(16, 128))
    with pytest.raises(RuntimeError):
        fr.get_min_valid_packet_timestamp()
    fr.status[:] = 1
    fr.status[16:32] = 0
    fr.packet_timestamp[:] = 50
    fr.packet_timestamp[1] = 1                     # packet 1 has no valid column
    fr.packet_timestamp[3] = 90
    assert (fr.get_min_valid_packet_timestamp(), fr.get_max_valid_packet_timestamp()) == (50, 90)
    assert fr.get_first_valid_packet_timestamp() == 50 and fr.get_last_valid_packet_timestamp() == 50
    fr.add_field("ZONE_PACKET_TIMESTAMP", np.array([95], np.uint64), core.FieldClass.FRAME_FIELD)
    assert fr.get_max_valid_packet_timestamp() == 95
    fr.add_field("IMU_PACKET_TIMESTAMP", np.array([3, 4], np.uint64), core.FieldClass.FRAME_FIELD)
    st = np.zeros((16,), np.uint16)
    st[9] = 1                                      # only the second IMU packet is valid
    fr.add_field("IMU_STATUS", st, core.FieldClass.FRAME_FIELD)
    assert fr.get_min_valid_packet_timestamp() == 4


def test_packet_headers_and_setter_bounds():
    pf = core.PacketFormat.from_info(_info())
    p = core.LidarPacket(pf)
    assert len(core.ColHeader.__members__) == 5
    for i in range(pf.columns_per_packet):
        pf.set_col_timestamp(p, i, 1000 + i)
        pf.set_col_measurement_id(p, i, 3 * i)
        pf.set_col_status(p, i, 1)
    assert list(pf.packet_header(core.ColHeader.TIMESTAMP, p.buf)) == [1000 + i for i in range(16)]
    assert list(pf.packet_header(core.ColHeader.MEASUREMENT_ID, p)) == [3 * i for i in range(16)]
    assert pf.packet_header(core.ColHeader.STATUS, p.buf).dtype == np.uint32
    for setter in (pf.set_col_timestamp, pf.set_col_measurement_id, pf.set_col_status):
        with pytest.raises(ValueError):
            setter(p, pf.columns_per_packet, 1)
    assert pf.shot_limiting(p.buf) == core.ShotLimitingStatus.NORMAL
    assert pf.thermal_shutdown(p.buf) == core.ThermalShutdownStatus.NORMAL
    assert core.PacketFormat(_info().format).lidar_packet_size == pf.lidar_packet_size
