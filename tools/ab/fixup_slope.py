#!/usr/bin/env python3
"""What does the fix-up pass cost per flagged frame?  256 dual-return frames, K of them with two packets swapped (strays),
K in {0, 8, 16, 32, 64, 128, 256}: ms per call -> slope and intercept."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ouster_sdk_amd.device import HotPath
wl = sys.argv[1] if len(sys.argv) > 1 else "dual"
H, W, N = bench.H, bench.W, 256
prof, bits, chan, dst, xyz = bench.WORKLOADS[wl][:5]
alt, az, shifts, b2l, l2s = bench.synth_calibration()
hp = HotPath(prof, H, W, 16)
hp.set_pixel_shift_by_row(shifts)
hp.add_lut(b2l, l2s, az, alt)
pool = bench.synth_packets(16, bits=bits, chan=chan)
base = torch.from_numpy(pool).repeat(N // 16, 1, 1).contiguous()
out = hp.alloc_outputs(N, destagger=dst, xyz=xyz)
res = {}
for K in (0, 8, 16, 32, 64, 128, 256):
    pk = base.clone()
    idx = np.linspace(0, N - 1, K).astype(int) if K else []
    for f in idx:
        pk[f, [10, 11]] = pk[f, [11, 10]]
    d = pk.cuda()
    for _ in range(30): hp.decode(d, out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(40): hp.decode(d, out)
    b.record(); torch.cuda.synchronize()
    res[K] = round(a.elapsed_time(b) / 40, 4)
ks = sorted(res)
slope = (res[ks[-1]] - res[ks[1]]) / (ks[-1] - ks[1]) * 1000
print(json.dumps({"workload": wl, "ms_per_call": res, "us_per_flagged_frame_8_to_256": round(slope, 2),
                  "first_8_frames_cost_us": round((res[8] - res[0]) * 1000, 1)}))
