#!/usr/bin/env python3
"""packets -> world-frame clouds: decode + two standalone dewarp passes against decode with xyz_poses (256 dual-return
frames of 128 x 2048, both returns)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ouster_sdk_amd.device import HotPath
H, W, N = bench.H, bench.W, 256
prof, bits, chan, dst, xyz = bench.WORKLOADS["dual"][:5]
alt, az, shifts, b2l, l2s = bench.synth_calibration()
hp = HotPath(prof, H, W, 16)
hp.set_pixel_shift_by_row(shifts)
hp.add_lut(b2l, l2s, az, alt)
pk = torch.from_numpy(bench.synth_packets(16, bits=bits, chan=chan)).cuda().repeat(N // 16, 1, 1).contiguous()
poses = torch.eye(4, dtype=torch.float64, device="cuda").repeat(N, W, 1, 1).contiguous()
poses[..., :3, 3] = torch.rand((N, W, 3), device="cuda", dtype=torch.float64) * 10
pk, out, _ = hp.pick_placement(pk, lambda: hp.alloc_outputs(N, destagger=dst, xyz=xyz), tries=12, stride_gb=4.0)
w1, w2 = torch.empty_like(out["xyz:RANGE"]), torch.empty_like(out["xyz:RANGE2"])
def timeit(fn, reps=20):
    for _ in range(16): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps, 4)
def two_pass():
    hp.decode(pk, out)
    hp.dewarp(out["xyz:RANGE"], poses); hp.dewarp(out["xyz:RANGE2"], poses)
res = {"decode_ms": timeit(lambda: hp.decode(pk, out)),
       "decode_with_poses_ms": timeit(lambda: hp.decode(pk, out, poses=poses)),
       "decode_then_two_dewarps_ms": timeit(two_pass)}
print(json.dumps(res))
