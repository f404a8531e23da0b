#!/usr/bin/env python3
"""HotPath.refine_placement on the first allocation of a fresh process: before / after, and what whole-set draws find.
usage: refine_check.py <workload> [draws] [passes]"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ouster_sdk_amd.device import HotPath
wl = sys.argv[1]
draws = int(sys.argv[2]) if len(sys.argv) > 2 else 3
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ballast = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
prof, bits, chan, dst, xyz = bench.WORKLOADS[wl][:5]
H, W, N = bench.H, bench.W, 256
alt, az, shifts, b2l, l2s = bench.synth_calibration()
pool = bench.synth_packets(16, bits=bits, chan=chan)
pk = torch.from_numpy(pool).cuda().repeat(N // 16, 1, 1).contiguous()
hp = HotPath(prof, H, W, 16)
hp.set_pixel_shift_by_row(shifts)
hp.add_lut(b2l, l2s, az, alt)
out = hp.alloc_outputs(N, destagger=dst, xyz=xyz)
for _ in range(20):
    hp.decode(pk, out)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
out, rep = hp.refine_placement(pk, out, draws=draws, ballast_gb=ballast)
rep["setup_s"] = round(time.perf_counter() - t0, 3)
rep["kernel"] = [hp.ctx.last_decode_kernel()] + list(hp.ctx.last_decode_tile())
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    hp.decode(pk, out)
b.record(); torch.cuda.synchronize()
rep["final_ms_50_calls"] = round(a.elapsed_time(b) / 50, 4)
print(json.dumps(rep))
