#!/usr/bin/env python3
"""K flagged frames of one kind, 20 calls: for rocprofv3 --kernel-trace --stats (what does the fix-up kernel itself take?)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
wl, kind, K = sys.argv[1], sys.argv[2], int(sys.argv[3])
N = 256
hp, base, out, *_ = bench._workload_setup(wl, N, pool_frames=8)
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    hp.ctx.set_knob(k, int(v))
base = base.cpu()
slots = base.shape[1]
pk = base.clone()
counts = torch.full((N,), slots, dtype=torch.int32)
for f in (np.linspace(0, N - 1, K).astype(int) if K else []):
    if kind == "swap":
        pk[f, [10, 11]] = pk[f, [11, 10]]
    elif kind == "swapshort":      # two packets swapped AND the last packet missing (count = slots - 1)
        pk[f, [10, 11]] = pk[f, [11, 10]]
        pk[f, slots - 1] = 0
        counts[f] = slots - 1
    elif kind == "rot":            # packets 20..29 rotated by one: ten strays in a row, count = slots
        pk[f, 20:30] = base[f, [29] + list(range(20, 29))]
    else:
        lost = int(os.environ.get('LOST', (int(f) * 7 + 3) % slots))
        keep = [p for p in range(slots) if p != lost]
        pk[f, :slots - 1] = base[f, keep]
        pk[f, slots - 1] = 0
        counts[f] = slots - 1
d, c = pk.cuda(), counts.cuda()
for _ in range(40):
    hp.decode(d, out, packet_counts=c)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(40):
    hp.decode(d, out, packet_counts=c)
b.record(); torch.cuda.synchronize()
print("us per call %.1f" % (a.elapsed_time(b) / 40 * 1000), flush=True)
