#!/usr/bin/env python3
"""packets -> planes (+ gate counts) -> world points: decode with / without the range-gate by-product,
dewarp with / without the counting pass (256 frames 128x2048 dual, gate 0.5-400 m)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ouster_sdk_amd.device import HotPath
H, W, N = 128, 2048, 256
alt, az, shifts, b2l, l2s = bench.synth_calibration()
hp = HotPath("RNG15_RFL8_NIR8_DUAL", H, W, 16)
hp.set_pixel_shift_by_row(shifts)
lut = hp.add_lut(b2l, l2s, az, alt)
pool = bench.synth_packets(16)
pk = torch.from_numpy(pool).cuda().repeat(N // 16, 1, 1).contiguous()
out = hp.alloc_outputs(N, destagger=bench.DESTAGGERED, xyz=["RANGE", "RANGE2"])
poses = torch.eye(4, dtype=torch.float64, device="cuda").repeat(N, W, 1, 1).contiguous()
gate = (0.5, 400.0)
def timeit(fn, reps=10):
    for _ in range(16): fn()   # the decode tuner settles after 12 calls
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
res = {}
res["decode_ms"] = timeit(lambda: hp.decode(pk, out))
res["decode_with_gate_counts_ms"] = timeit(lambda: hp.decode(pk, out, gate=gate))
kept = None
for tag in ("runs",):
    res[f"dewarp_{tag}_ms"] = timeit(lambda: hp.dewarp_frames(out["RANGE"], out["status"], poses, *gate, provenance=False, luts=[lut]))
    res[f"dewarp_{tag}_counts_from_decode_ms"] = timeit(lambda: hp.dewarp_frames(out["RANGE"], out["status"], poses, *gate, provenance=False, luts=[lut], gate_counts=out["gate_counts"]))
    res[f"dewarp_{tag}_provenance_ms"] = timeit(lambda: hp.dewarp_frames(out["RANGE"], out["status"], poses, *gate, timestamp=out["timestamp"], luts=[lut], gate_counts=out["gate_counts"]))
    got = hp.dewarp_frames(out["RANGE"], out["status"], poses, *gate, provenance=False, luts=[lut], gate_counts=out["gate_counts"])
    kept = int(got["frame_offsets"][-1])
    chk = float(got["points"][:kept].double().sum())
    res[f"checksum_{tag}"] = chk
res["kept_points"] = kept
res["algorithmic_MB"] = round((N * H * W * 4 + kept * 12) / 1e6, 1)
print(json.dumps({k: round(v, 4) for k, v in res.items()}))
