"""Sensor metadata JSON -> `core.SensorInfo` (the calibration and data-format subset the hot path needs).

Both generations of the file are read, as the reference does (ouster_core/src/metadata.cpp:831-842 rewrites the legacy
flat keys into the nested form, :482-592 parses the data format, :725-771 the intrinsics):
  * nested: beam_intrinsics.*, lidar_data_format.*, lidar_intrinsics.lidar_to_sensor_transform, sensor_info.*,
            config_params.{lidar_mode, udp_profile_lidar}, optional 'ouster-sdk'.extrinsic;
  * legacy flat: beam_altitude_angles, beam_azimuth_angles, data_format.*, lidar_to_sensor_transform,
            lidar_origin_to_beam_origin_mm, prod_line, prod_sn, initialization_id, build_rev, lidar_mode.
Defaults follow default_data_format(mode) (data_format.cpp:79-125: 64 px, 16 columns per packet, LEGACY profile,
shifts {18, 12, 6, 0} * W / 1024 x 16), pixel_shift_by_row is zero-padded / cut to H (metadata.cpp:530-534), a
FUSA_RNG15_RFL8_NIR8_DUAL profile without header_type implies FUSA headers (:545-555), a missing
beam_to_lidar_transform is identity with (0, 3) = lidar_origin_to_beam_origin_mm (sensor_info.cpp:89-105).
Everything else of the reference's metadata handling (config, calibration status, zone sets, ...) is out of scope.
"""
import json
from typing import Any, Dict

import numpy as np


def sensor_info_from_json(text: str):
    from . import core
    d: Dict[str, Any] = json.loads(text)
    nested = "lidar_data_format" in d or "beam_intrinsics" in d
    if nested:
        df = d.get("lidar_data_format", {})
        bi = d.get("beam_intrinsics", {})
        si = d.get("sensor_info", {})
        l2s = d.get("lidar_intrinsics", {}).get("lidar_to_sensor_transform")
        cfg = d.get("config_params", {})
        header_type = df.get("header_type") or cfg.get("header_type")   # metadata.cpp:545-555 reads lidar_data_format first
    else:
        df, bi, si, cfg = d.get("data_format", {}), d, d, d
        l2s = d.get("lidar_to_sensor_transform")
        header_type = None
    mode = cfg.get("lidar_mode") or d.get("lidar_mode") or "1024x10"
    w_mode, fps = (int(x) for x in mode.split("x")[:2]) if "x" in mode else (1024, 10)
    info = core.SensorInfo()
    f = info.format
    f.pixels_per_column = int(df.get("pixels_per_column", 64))
    f.columns_per_packet = int(df.get("columns_per_packet", 16))
    f.columns_per_frame = int(df.get("columns_per_frame", w_mode))
    f.fps = int(df.get("fps", fps))
    h, w = f.pixels_per_column, f.columns_per_frame
    shifts = df.get("pixel_shift_by_row")
    if shifts is None:
        shifts = [x * w // 1024 for x in (18, 12, 6, 0)] * 16
    f.pixel_shift_by_row = (list(int(x) for x in shifts) + [0] * h)[:h]
    cw = df.get("column_window", [0, w - 1])
    f.column_window = (int(cw[0]), int(cw[1]))
    profile = df.get("udp_profile_lidar") or cfg.get("udp_profile_lidar") or "LEGACY"
    f.udp_profile_lidar = core.UDPProfileLidar.from_string(profile)
    if header_type is None:
        header_type = "FUSA" if profile == "FUSA_RNG15_RFL8_NIR8_DUAL" else "STANDARD"
    f.header_type = core.HeaderType.FUSA if str(header_type).upper() == "FUSA" else core.HeaderType.STANDARD
    info.format = f
    info.beam_altitude_angles = [float(x) for x in np.asarray(bi.get("beam_altitude_angles", [])).reshape(-1)]
    info.beam_azimuth_angles = [float(x) for x in np.asarray(bi.get("beam_azimuth_angles", [])).reshape(-1)]
    info.prod_line = str(si.get("prod_line", ""))
    origin = bi.get("lidar_origin_to_beam_origin_mm")
    b2l = bi.get("beam_to_lidar_transform")
    if b2l is not None:
        b2l = np.asarray(b2l, dtype=np.float64).reshape(4, 4)
        origin = float(b2l[0, 3]) if origin is None else float(origin)
    else:
        if origin is None:
            origin = float(core.default_beam_to_lidar_transform(info.prod_line)[0, 3])
        b2l = np.eye(4)
        b2l[0, 3] = float(origin)
    info.lidar_origin_to_beam_origin_mm = float(origin)
    info.beam_to_lidar_transform = b2l
    info.lidar_to_sensor_transform = (np.asarray(l2s, dtype=np.float64).reshape(4, 4) if l2s
                                      else core.default_lidar_to_sensor())
    ext = d.get("ouster-sdk", {}).get("extrinsic") if isinstance(d.get("ouster-sdk"), dict) else None
    info.sensor_to_body = np.asarray(ext, dtype=np.float64).reshape(4, 4) if ext else np.eye(4)
    info.init_id = int(si.get("initialization_id", 0) or 0)
    try:
        info.sn = int(si.get("prod_sn", 0) or 0)
    except ValueError:
        info.sn = 0
    info.fw_rev = str(si.get("build_rev", ""))
    return info
