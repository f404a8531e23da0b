#!/usr/bin/env python3
"""alloc_lottery.py arranged for a rocprofv3 --pmc pass: K output sets, L launches into each, in order; prints how
many decode launches precede the sets so that the counter rows can be mapped to the sets by dispatch order."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ouster_sdk_amd.device import HotPath
K, L = 8, 6
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
def slab_set():
    tmpl = hp.alloc_outputs(N, destagger=dst, xyz=xyz)
    names = list(tmpl); sizes = [tmpl[n].numel() * tmpl[n].element_size() for n in names]
    meta = {n: (tmpl[n].dtype, tuple(tmpl[n].shape)) for n in names}
    del tmpl; torch.cuda.empty_cache()
    slab = torch.empty(sum((x + al - 1) // al * al for x in sizes) + al, dtype=torch.uint8, device="cuda")
    off = (-slab.data_ptr()) % al
    out = {}
    for n, nb in zip(names, sizes):
        out[n] = slab[off:off + nb].view(meta[n][0]).view(meta[n][1]); off += (nb + al - 1) // al * al
    return out
sets = [slab_set() for _ in range(K)]
torch.cuda.synchronize()
ms = []
for o in sets:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(L): hp.decode(pk, o)
    b.record(); torch.cuda.synchronize()
    ms.append(round(a.elapsed_time(b) / L, 4))
print(json.dumps({"sets": K, "launches_per_set": L, "ms_per_launch_under_the_profiler": ms}))
