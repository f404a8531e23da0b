"""The C++ metadata reader (csrc/host/metadata.cpp: SensorInfo(json_text), metadata_from_json) against the Python one
(ouster_sdk_amd/metadata.py) on every metadata file among the fixtures -- both generations of the file (flat legacy keys,
nested current ones) -- and on malformed input.  No GPU involved."""
import glob
import os

import numpy as np
import pytest

from conftest import PCAPS


def _files():
    return [f for f in sorted(glob.glob(os.path.join(PCAPS, "*.json"))) if not f.endswith("_digest.json")]


@pytest.mark.parametrize("path", _files(), ids=os.path.basename)
def test_cpp_reader_equals_python_reader(path):
    from ouster_sdk_amd import core
    from ouster_sdk_amd.metadata import sensor_info_from_json
    text = open(path).read()
    a, b = core.SensorInfo(text), sensor_info_from_json(text)
    for k in ("sn", "init_id", "prod_line", "lidar_origin_to_beam_origin_mm", "beam_altitude_angles", "beam_azimuth_angles"):
        assert getattr(a, k) == getattr(b, k), k
    for k in ("pixels_per_column", "columns_per_packet", "columns_per_frame", "pixel_shift_by_row", "column_window",
              "udp_profile_lidar", "header_type", "fps"):
        assert getattr(a.format, k) == getattr(b.format, k), k
    for k in ("beam_to_lidar_transform", "lidar_to_sensor_transform", "sensor_to_body"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    assert a.config.lidar_mode is None or a.config.lidar_mode.columns in (512, 1024, 2048, 4096)


@pytest.mark.parametrize("bad", ["", "{", "[" * 200, '{"a": 1,}', '{"a": "unterminated',
                                 '{"lidar_data_format": {"udp_profile_lidar": "NO_SUCH_PROFILE"}}'])
def test_malformed_metadata_is_refused(bad):
    from ouster_sdk_amd import core
    with pytest.raises(RuntimeError):
        core.SensorInfo(bad)
