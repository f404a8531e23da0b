#!/usr/bin/env python3
"""How much of the box-to-box / process-to-process spread is the placement of THIS process's output planes?
One process, one library, one packet buffer: K complete output sets are allocated (all kept alive) and the
decode is timed into each of them in alternating blocks."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ouster_sdk_amd.device import HotPath
wl = sys.argv[1] if len(sys.argv) > 1 else "dual"
wide = int(sys.argv[2]) if len(sys.argv) > 2 else 256
K = int(sys.argv[3]) if len(sys.argv) > 3 else 6
prof, bits, chan, dst, xyz = bench.WORKLOADS[wl][:5]
H, W, N = bench.H, bench.W, 256
alt, az, shifts, b2l, l2s = bench.synth_calibration()
pool = bench.synth_packets(16, bits=bits, chan=chan)
pk = torch.from_numpy(pool).cuda().repeat(N // 16, 1, 1).contiguous()
hp = HotPath(prof, H, W, 16)
hp.set_pixel_shift_by_row(shifts)
hp.add_lut(b2l, l2s, az, alt)
hp.ctx.set_knob("wide", wide)
sets = [hp.alloc_outputs(N, destagger=dst, xyz=xyz) for _ in range(K)]
pks = [pk] + [pk.clone() for _ in range(2)]
for o in sets:
    for _ in range(3):
        hp.decode(pk, o)
torch.cuda.synchronize()
def t(pkb, o):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    hp.decode(pkb, o); a.record()
    for _ in range(20): hp.decode(pkb, o)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 20
res = {"workload": wl, "wide": wide, "output_sets_ms": [], "input_copies_ms": []}
for i, o in enumerate(sets):
    res["output_sets_ms"].append(round(float(np.median([t(pk, o) for _ in range(4)])), 4))
# optional: the same sets under another knob setting (argv[4] = knob:value), e.g. xcd:2
if len(sys.argv) > 4:
    k, v = sys.argv[4].split(":")
    hp.ctx.set_knob(k, int(v))
    res["output_sets_ms_" + sys.argv[4]] = [round(float(np.median([t(pk, o) for _ in range(4)])), 4) for o in sets]
    hp.ctx.set_knob(k, 1)
for pkb in pks:
    res["input_copies_ms"].append(round(float(np.median([t(pkb, sets[0]) for _ in range(4)])), 4))
print(json.dumps(res))
