#!/usr/bin/env python3
"""In-process A/B of library builds on the standalone kernels (destagger, cartesian, dense dewarp, frame dewarp):
same input and output tensors for every build.  usage: ab_standalone.py name=path.so ..."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ouster_sdk_amd import _capi
from ouster_sdk_amd.device import HotPath
H, W, N = 128, 2048, 256
alt, az, shifts, b2l, l2s = bench.synth_calibration()
rng = torch.randint(0, 2 ** 18, (N, H, W), dtype=torch.int64, device="cuda")
rng[torch.rand((N, H, W), device="cuda") < 0.3] = 0
rng = rng.to(torch.uint32)
status = torch.ones((N, W), dtype=torch.uint32, device="cuda")
poses = torch.eye(4, dtype=torch.float64, device="cuda").repeat(N, W, 1, 1).contiguous()
pts = torch.randn((64, H * W, 3), dtype=torch.float32, device="cuda")
poses64 = poses[:64].contiguous()
ts0 = torch.zeros((N, W), dtype=torch.int64, device="cuda").to(torch.uint64)
variants = [a.split("=", 1) for a in sys.argv[1:]]
hps = {}
for name, path in variants:
    lib = _capi.load_hip(os.path.join(ROOT, path)) if path else None
    hp = HotPath("RNG15_RFL8_NIR8_DUAL", H, W, 16, lib=lib)
    hp.set_pixel_shift_by_row(shifts)
    hp.add_lut(b2l, l2s, az, alt)
    hps[name] = hp
# four copies of every input, used in turn: no call finds its input in the 256 MB Infinity Cache
R = 4
rngs = [rng] + [rng.clone() for _ in range(R - 1)]
ptss = [pts] + [pts.clone() for _ in range(R - 1)]
turn = [0]
def nxt(lst):
    turn[0] += 1
    return lst[turn[0] % R]
ops = {
    "destagger_u32": lambda hp: hp.destagger(nxt(rngs)),
    "cartesian_f32": lambda hp: hp.cartesian(nxt(rngs), dtype=torch.float32),
    "cartesian_f64": lambda hp: hp.cartesian(nxt(rngs), dtype=torch.float64),
    "dewarp_f32": lambda hp: hp.dewarp(nxt(ptss), poses64),
    "dewarp_frames_f32": lambda hp: hp.dewarp_frames(nxt(rngs), status, poses, 0.5, 400.0, provenance=False),
    "dewarp_frames_f32_provenance": lambda hp: hp.dewarp_frames(nxt(rngs), status, poses, 0.5, 400.0, timestamp=ts0, provenance=True),
}
if os.environ.get("AB_OPS"):
    ops = {k: v for k, v in ops.items() if k in os.environ["AB_OPS"].split(",")}
res = {}
for op, fn in ops.items():
    times = {n: [] for n in hps}
    for rnd in range(5):
        for name, hp in hps.items():
            fn(hp); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                fn(hp)
            b.record(); torch.cuda.synchronize()
            times[name].append(a.elapsed_time(b) / 10)
    res[op] = {n: round(float(np.median(t)), 4) for n, t in times.items()}
print(json.dumps(res))
