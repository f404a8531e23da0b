#!/usr/bin/env python3
"""Standalone-kernel rooflines (k_destagger, k_cartesian, k_dewarp) on resident data.
Prints one JSON object; algorithmic bytes per the definitions in DESIGN.md section 3."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ouster_sdk_amd.device import HotPath

H, W, N = 128, 2048, 256


R = 4   # copies of every input, used in turn: a call must not find its input in the 256 MB Infinity Cache (with the
        # kernels' non-temporal stores a re-read input survives there and inflates the rates by 20-40 %)


class Rot:
    def __init__(self, t):
        self.c, self.i = [t] + [t.clone() for _ in range(R - 1)], 0

    def __call__(self):
        self.i += 1
        return self.c[self.i % R]


def timeit(fn, reps=12):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def main():
    alt, az, shifts, b2l, l2s = bench.synth_calibration()
    hp = HotPath("RNG15_RFL8_NIR8_DUAL", H, W, 16)
    hp.set_pixel_shift_by_row(shifts)
    lut = hp.add_lut(b2l, l2s, az, alt)
    d, o = lut.export(W, H)
    lut32 = hp.add_lut_arrays(d.astype(np.float32), o.astype(np.float32))
    rng = torch.randint(0, 2 ** 19, (N, H, W), dtype=torch.int64, device="cuda").to(torch.uint32)
    res = {}
    for name, t in (("u32", rng), ("u8", rng.to(torch.uint8)), ("u16", rng.to(torch.uint16))):
        rt = Rot(t)
        s = timeit(lambda: hp.destagger(rt()))
        res[f"destagger_{name}"] = {"GBps": round(2 * t.numel() * t.element_size() / s / 1e9, 1),
                                    "ms": round(s * 1e3, 3)}
    npx = N * H * W
    for name, l, dt, bpp in (("sep_f32", lut, torch.float32, 4 + 12), ("sep_f64", lut, torch.float64, 4 + 24),
                             ("fullLUT_f32", lut32, torch.float32, 4 + 12 + 24 / N)):  # the LUT is shared by the N images
        rr = Rot(rng)
        s = timeit(lambda: hp.cartesian(rr(), lut=l, dtype=dt))
        res[f"cartesian_{name}"] = {"GBps": round(npx * bpp / s / 1e9, 1), "Mpoints_per_s": round(npx / s / 1e6, 1),
                                    "ms": round(s * 1e3, 3)}
    pts = hp.cartesian(rng[:64].contiguous())
    poses = torch.eye(4, dtype=torch.float64, device="cuda").repeat(64, W, 1, 1).contiguous()
    rp = Rot(pts)
    s = timeit(lambda: hp.dewarp(rp(), poses))
    res["dewarp_f32"] = {"GBps": round(2 * pts.numel() * 4 / s / 1e9, 1),
                         "Mpoints_per_s": round(pts.numel() / 3 / s / 1e6, 1), "ms": round(s * 1e3, 3)}
    pts64 = pts.double()
    rp64 = Rot(pts64)
    s = timeit(lambda: hp.dewarp(rp64(), poses))
    res["dewarp_f64"] = {"GBps": round(2 * pts64.numel() * 8 / s / 1e9, 1),
                         "Mpoints_per_s": round(pts64.numel() / 3 / s / 1e6, 1), "ms": round(s * 1e3, 3)}
    # range-gated compacting dewarp of whole frames (dewarp_impl.h:23-115): ~30 % zeros in the ranges
    rz = (rng.to(torch.int64) * (torch.rand(rng.shape, device="cuda") >= 0.3)).to(torch.uint32)
    status = torch.ones((N, W), dtype=torch.int32, device="cuda").to(torch.uint32)
    ts = torch.arange(N * W, dtype=torch.int64, device="cuda").reshape(N, W).to(torch.uint64)
    posesN = torch.eye(4, dtype=torch.float64, device="cuda").repeat(N, W, 1, 1).contiguous()
    for name, prov in (("dewarp_frames_f32", False), ("dewarp_frames_f32_provenance", True)):
        out = hp.dewarp_frames(rz, status, posesN, 0.5, 400.0, timestamp=ts if prov else None, provenance=prov, luts=[lut])
        kept = int(out["frame_offsets"][-1].item())
        rrz = Rot(rz)
        s = timeit(lambda: hp.dewarp_frames(rrz(), status, posesN, 0.5, 400.0, timestamp=ts if prov else None, luts=[lut],
                                            provenance=prov))
        byts = npx * 4 + kept * (12 + (16 if prov else 0)) + N * W * (128 + 4)
        res[name] = {"GBps": round(byts / s / 1e9, 1), "Mpixels_per_s": round(npx / s / 1e6, 1),
                     "kept_fraction": round(kept / npx, 3), "ms": round(s * 1e3, 3)}
    # the same with the poses as float rows (ouster_hip_dewarp_frames_rows: what a caller that stages poses on the host hands
    # over: 48 B per column) and with the per-column counts the decode leaves behind when it runs with the gate
    # (ouster_hip_frame_out::gate_counts: no counting pass over the range planes) -- DeviceFrameBatch::dewarp's route
    rowsN = HotPath.pose_rows(posesN)
    hp2 = HotPath("RNG15_RFL8_NIR8_DUAL", H, W, 16)
    hp2.set_pixel_shift_by_row(shifts)
    hp2.add_lut(b2l, l2s, az, alt)
    pk = torch.from_numpy(bench.synth_packets(8)).cuda().repeat(N // 8, 1, 1).contiguous()
    dout = hp2.alloc_outputs(N, planes=["RANGE"], xyz=[])
    hp2.decode(pk, dout, gate=(0.5, 400.0))
    drng, dst_, dgc = dout["RANGE"], dout["status"], dout["gate_counts"]
    for name, kw in (("dewarp_frames_f32_rows", dict(rng=rz, st=status, gc=None)),
                     ("dewarp_frames_f32_rows_counted", dict(rng=drng, st=dst_, gc=dgc))):
        out = hp.dewarp_frames(kw["rng"], kw["st"], rowsN, 0.5, 400.0, provenance=False, luts=[lut], gate_counts=kw["gc"])
        kept = int(out["frame_offsets"][-1].item())
        rr_ = Rot(kw["rng"])
        s = timeit(lambda: hp.dewarp_frames(rr_(), kw["st"], rowsN, 0.5, 400.0, luts=[lut], provenance=False, gate_counts=kw["gc"]))
        byts = npx * 4 + kept * 12 + N * W * (48 + 4) + (N * W * 2 * 8 if kw["gc"] is not None else 0)
        res[name] = {"GBps": round(byts / s / 1e9, 1), "Mpixels_per_s": round(npx / s / 1e6, 1),
                     "kept_fraction": round(kept / npx, 3), "ms": round(s * 1e3, 3), "algorithmic_bytes": int(byts)}
    res["note"] = (f"{N} images of {H}x{W}; {R} copies of every input used in turn (cold input for every call); "
                   "includes torch.empty_like of the output per call")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
