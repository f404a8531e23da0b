#!/usr/bin/env python3
"""OSF LidarScan messages -> planes in HBM (OsfFrameDecoder.decode_device) with the PNG scanline filters reversed on the GPU
(k_osf_png_unfilter, the default since round 5) and on the host (rounds 2 - 4): milliseconds per batch of 96 messages of the
reference's fixtures.  The host half (zlib inflate on a parked crew of up to 128 threads, straight into one pinned buffer)
dominates either way."""
import sys, time, json, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import test_gpu_osf as T
from oracle import osf_oracle as Z
from ouster_sdk_amd import core
res = {}
for label, path in (("png16_lb", T.LB), ("png8", T.PNG8)):
    zf = Z.OsfFile(path)
    meta = list(zf.sensor_metadata().values())[0]
    pf = core.OsfFile(path)
    streams = pf.lidar_scan_streams()
    msgs = [m for (_, sid, m) in pf.messages() if sid in streams]
    batch = (msgs * 64)[:96]
    info = T._sensor_info(core, meta)
    r = {"frames": len(batch), "h_w": T._geometry(meta)[:2]}
    for on in (True, False):
        dec = core.OsfFrameDecoder(info)
        dec.device_unfilter = on
        dec.decode_device(batch[:8])
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); b = dec.decode_device(batch); best = min(best, time.perf_counter() - t0)
        r["device_unfilter" if on else "host_unfilter"] = round(best * 1e3, 2)
        if on:   # the same batch back on the host as LidarFrames (OsfFrameDecoder.decode)
            dec.decode(batch[:8])
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); fr = dec.decode(batch); best = min(best, time.perf_counter() - t0)
            r["decode_to_host_frames"] = round(best * 1e3, 2)
    res[label] = r
print(json.dumps(res))
