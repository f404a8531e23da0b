#!/usr/bin/env python3
"""dewarp_frames only (for rocprofv3 passes): 256 frames 128x2048, 30 % zero ranges, gate 0.5-400 m."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ouster_sdk_amd.device import HotPath
H, W, N = 128, 2048, 256
alt, az, shifts, b2l, l2s = bench.synth_calibration()
hp = HotPath("RNG15_RFL8_NIR8_DUAL", H, W, 16)
lut = hp.add_lut(b2l, l2s, az, alt)
g = torch.Generator(device="cuda").manual_seed(1)
rz = torch.randint(0, 2 ** 19, (N, H, W), dtype=torch.int64, device="cuda", generator=g)
rz[torch.rand((N, H, W), device="cuda", generator=g) < 0.3] = 0
rz = rz.to(torch.uint32)
status = torch.ones((N, W), dtype=torch.int32, device="cuda").to(torch.uint32)
poses = torch.eye(4, dtype=torch.float64, device="cuda").repeat(N, W, 1, 1).contiguous()
prov = len(sys.argv) > 1 and sys.argv[1] == "prov"
ts = torch.zeros((N, W), dtype=torch.int64, device="cuda").to(torch.uint64) if prov else None
reps = int(os.environ.get("REPS", "5"))
for _ in range(reps):
    out = hp.dewarp_frames(rz, status, poses, 0.5, 400.0, timestamp=ts, provenance=prov, luts=[lut])
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    out = hp.dewarp_frames(rz, status, poses, 0.5, 400.0, timestamp=ts, provenance=prov, luts=[lut])
b.record(); torch.cuda.synchronize()
print(json.dumps({"ms": a.elapsed_time(b) / reps, "kept": int(out["frame_offsets"][-1].item()) / (N * H * W)}))
