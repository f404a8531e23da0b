#!/usr/bin/env python3
"""Placement x kernel-variant matrix: K complete output sets are allocated side by side in ONE process (every set a
different draw of the allocation lottery; optional ballast between them so that they scan the device
memory) and the same decode is timed into each of them with every variant.
usage: lottery_variants.py <workload> <K> <ballast GB> name=knob:value[,knob:value...] ..."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ouster_sdk_amd.device import HotPath

wl, K, ballast_gb = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
variants = []
for a in sys.argv[4:]:
    name, spec = a.split("=", 1)
    variants.append((name, [kv.split(":") for kv in spec.split(",") if kv]))
prof, bits, chan, dst, xyz = bench.WORKLOADS[wl][:5]
H, W, N = bench.H, bench.W, 256
alt, az, shifts, b2l, l2s = bench.synth_calibration()
pool = bench.synth_packets(16, bits=bits, chan=chan)
pk = torch.from_numpy(pool).cuda().repeat(N // 16, 1, 1).contiguous()
hps = {}
for name, knobs in variants:
    hp = HotPath(prof, H, W, 16)
    hp.set_pixel_shift_by_row(shifts)
    hp.add_lut(b2l, l2s, az, alt)
    hp.ctx.set_knob("tune", 0)
    for k, v in knobs:
        hp.ctx.set_knob(k, int(v))
    hps[name] = hp
first = next(iter(hps.values()))
sets, held = [], []
for _ in range(K):
    sets.append(first.alloc_outputs(N, destagger=dst, xyz=xyz))
    if ballast_gb > 0:
        held.append(torch.empty(int(ballast_gb * (1 << 30)), dtype=torch.uint8, device="cuda"))


def t(hp, o):
    for _ in range(2):
        hp.decode(pk, o)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(12):
        hp.decode(pk, o)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 12


res = {"workload": wl, "sets": K, "ballast_gb": ballast_gb, "ms": {n: [] for n in hps},
       "kernels": {}}
for o in sets:
    for name, hp in hps.items():
        res["ms"][name].append(round(float(np.median([t(hp, o) for _ in range(3)])), 4))
for name, hp in hps.items():
    res["kernels"][name] = [hp.ctx.last_decode_kernel()] + list(hp.ctx.last_decode_tile())
res["best_of_variants_per_set"] = [min(res["ms"][n][i] for n in hps) for i in range(K)]
print(json.dumps(res))
