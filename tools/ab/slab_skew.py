#!/usr/bin/env python3
"""Is the allocation lottery (tools/ab/alloc_lottery.py) about how the planes sit RELATIVE to each other?
All outputs are carved out of slabs with controlled offsets between the planes; several independent slabs of
each layout, everything in one process."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ouster_sdk_amd.device import HotPath
wl = sys.argv[1] if len(sys.argv) > 1 else "dual"
wide = int(sys.argv[2]) if len(sys.argv) > 2 else 256
prof, bits, chan, dst, xyz = bench.WORKLOADS[wl][:5]
H, W, N = bench.H, bench.W, 256
alt, az, shifts, b2l, l2s = bench.synth_calibration()
pool = bench.synth_packets(16, bits=bits, chan=chan)
pk = torch.from_numpy(pool).cuda().repeat(N // 16, 1, 1).contiguous()
hp = HotPath(prof, H, W, 16)
hp.set_pixel_shift_by_row(shifts)
hp.add_lut(b2l, l2s, az, alt)
hp.ctx.set_knob("wide", wide)
tmpl = hp.alloc_outputs(N, destagger=dst, xyz=xyz)
names = list(tmpl)
MB2 = 2 << 20
def carve(skew_of):
    sizes = [tmpl[n].numel() * tmpl[n].element_size() for n in names]
    total = sum((s + MB2 - 1) // MB2 * MB2 + MB2 for s in sizes) + (200 << 20)
    slab = torch.empty(total, dtype=torch.uint8, device="cuda")
    base = (-slab.data_ptr()) % MB2
    out, off = {}, base
    for k, n in enumerate(names):
        o = off + skew_of(k)
        nb = sizes[k]
        out[n] = slab[o:o + nb].view(tmpl[n].dtype).view(tmpl[n].shape)
        off += (nb + MB2 - 1) // MB2 * MB2 + MB2
    out["_slab"] = slab
    return out
layouts = {
    "aligned_2MB": lambda k: 0,
    "skew_2MB": lambda k: (2 << 20) * k,
    "skew_6MB": lambda k: (6 << 20) * k,
    "skew_2MB_mod8": lambda k: (2 << 20) * ((k * 3) % 8),
    "skew_1MB+4KB": lambda k: ((1 << 20) + 4096) * k,
    "skew_256B": lambda k: 256 * k,
    "skew_4352B": lambda k: 4352 * k,
    "skew_64KB+256": lambda k: (65536 + 256) * k,
    "skew_136KB": lambda k: (128 * 1024 + 8192 + 512) * k,
}
def t(o):
    oo = {k: v for k, v in o.items() if k != "_slab"}
    for _ in range(3): hp.decode(pk, oo)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): hp.decode(pk, oo)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 20
res = {"workload": wl, "wide": wide, "separate_allocations_ms": round(t(tmpl), 4)}
for name, fn in layouts.items():
    sets = [carve(fn) for _ in range(3)]
    res[name] = [round(float(np.median([t(o) for _ in range(3)])), 4) for o in sets]
    del sets
    torch.cuda.empty_cache()
print(json.dumps(res))
