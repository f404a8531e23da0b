"""What a caller of the Python drop-in (`ouster_sdk_amd.core`, the reference's `ouster.sdk.core` call shapes) pays per
128 x 2048 dual-return frame: FrameBatcher over 128 packets, destagger of RANGE, XYZLut() -- numpy in, numpy out.  The C++
figures of the same calls: tools/bench_host_api.cpp.  One JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ouster_sdk_amd import core   # noqa: E402

H, W, CPP = 128, 2048, 16
info = core.SensorInfo()
f = info.format
f.pixels_per_column, f.columns_per_frame, f.columns_per_packet = H, W, CPP
f.column_window = (0, W - 1)
f.udp_profile_lidar = core.UDPProfileLidar.from_string("RNG15_RFL8_NIR8_DUAL")
f.pixel_shift_by_row = [(24, 8, -8, -24)[i % 4] for i in range(H)]
info.format = f
info.beam_azimuth_angles = [(4.2, 1.4, -1.4, -4.2)[i % 4] for i in range(H)]
info.beam_altitude_angles = [21.0 - 42.0 * i / (H - 1) for i in range(H)]
b2l = np.eye(4); b2l[0, 3] = 13.762
info.beam_to_lidar_transform = b2l
l2s = np.diag([-1.0, -1.0, 1.0, 1.0]); l2s[2, 3] = 36.18
info.lidar_to_sensor_transform = l2s
info.sensor_to_body = np.eye(4)
info.init_id = 77
info.fw_rev = "v3.2.0"
info.prod_line = "OS-2-128"

pf = core.PacketFormat(info)
src = core.LidarFrame(info)
rng = np.random.default_rng(1)
for name in ("RANGE", "RANGE2"):
    src.field(name)[:] = (rng.integers(0, 1 << 15, size=(H, W)).astype(np.uint32) << 3)
src.field("REFLECTIVITY")[:] = rng.integers(0, 256, size=(H, W)).astype(np.uint8)
src.status[:] = 1
src.measurement_id[:] = np.arange(W, dtype=np.uint16)
src.timestamp[:] = np.arange(W, dtype=np.uint64)
src.packet_timestamp[:] = 1 + np.arange(W // CPP, dtype=np.uint64)
lut = core.XYZLut(info, False)
frame = core.LidarFrame(info)
batcher = core.FrameBatcher(info)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tb = td = tx = 0.0
ok = True
for it in range(-3, N):
    src.frame_id = 100 + it + 3
    packets = core.frame_to_packets(src, pf, info.init_id, 1)
    t0 = time.perf_counter()
    done = False
    for p in packets:
        done = batcher(p, frame)
    t1 = time.perf_counter()
    d = core.destagger(info, frame.field("RANGE"))
    t2 = time.perf_counter()
    xyz = lut(frame)
    t3 = time.perf_counter()
    ok = ok and done and d.shape == (H, W) and xyz.shape == (H, W, 3)
    if it >= 0:
        tb += t1 - t0; td += t2 - t1; tx += t3 - t2
ok = ok and bool(np.array_equal(frame.field("RANGE"), src.field("RANGE")))
ok = ok and bool(np.array_equal(d, np.stack([np.roll(r, s) for r, s in zip(frame.field("RANGE"), f.pixel_shift_by_row)])))
print(json.dumps({"frames": N, "ms_per_frame": {"FrameBatcher_128_packets": round(tb / N * 1e3, 4), "destagger_u32": round(td / N * 1e3, 4),
                                                "XYZLut_f64": round(tx / N * 1e3, 4)}, "ok": ok,
                  "what": "ouster_sdk_amd.core (pybind11): numpy in, numpy out"}))
