#!/usr/bin/env python3
"""Does WHERE in the device memory an allocation lands decide its mode (alloc_lottery.py)?  Hold B GB of ballast,
then allocate output slabs and time the decode into them; B = 0 ... 200 GB, one process."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ouster_sdk_amd.device import HotPath
prof, bits, chan, dst, xyz = bench.WORKLOADS["dual"][:5]
H, W, N = bench.H, bench.W, 256
alt, az, shifts, b2l, l2s = bench.synth_calibration()
pool = bench.synth_packets(16, bits=bits, chan=chan)
pk = torch.from_numpy(pool).cuda().repeat(N // 16, 1, 1).contiguous()
hp = HotPath(prof, H, W, 16)
hp.set_pixel_shift_by_row(shifts)
hp.add_lut(b2l, l2s, az, alt)
hp.ctx.set_knob("wide", 256)
al = 2 << 20
tmpl = hp.alloc_outputs(N, destagger=dst, xyz=xyz)
names = list(tmpl); sizes = [tmpl[n].numel() * tmpl[n].element_size() for n in names]
meta = {n: (tmpl[n].dtype, tuple(tmpl[n].shape)) for n in names}
for _ in range(14): hp.decode(pk, tmpl)
del tmpl; torch.cuda.empty_cache()
def slab_set():
    slab = torch.empty(sum((x + al - 1) // al * al for x in sizes) + al, dtype=torch.uint8, device="cuda")
    off = (-slab.data_ptr()) % al
    out = {}
    for n, nb in zip(names, sizes):
        out[n] = slab[off:off + nb].view(meta[n][0]).view(meta[n][1]); off += (nb + al - 1) // al * al
    return out
def t(o):
    for _ in range(3): hp.decode(pk, o)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(12): hp.decode(pk, o)
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / 12, 4)
res = {}
ballast = []
held = 0
for target in (0, 8, 16, 24, 32, 48, 64, 96, 128, 160, 200):
    while held < target:
        ballast.append(torch.empty(4 << 30, dtype=torch.uint8, device="cuda")); held += 4
    sets = [slab_set() for _ in range(3)]
    res[f"{target}GB"] = [t(o) for o in sets]
    del sets; torch.cuda.empty_cache()
    print(json.dumps({f"{target}GB_ballast": res[f"{target}GB"]}), flush=True)
