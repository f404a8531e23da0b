#!/usr/bin/env python3
"""Fix-up pass cost by kind of damage: K frames with two packets swapped / compacted after a drop, K in {0,1,2,4,8,16,32,64}."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "dual"
N = 256
hp, base, out, *_ = bench._workload_setup(wl, N, pool_frames=8)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    hp.ctx.set_knob(k, int(v))
base = base.cpu()
slots = base.shape[1]
for kind in ("swap", "compact"):
    res = {}
    for K in (0, 1, 2, 4, 8, 16, 32, 64):
        pk = base.clone()
        counts = torch.full((N,), slots, dtype=torch.int32)
        idx = np.linspace(0, N - 1, K).astype(int) if K else []
        for f in idx:
            if kind == "swap":
                pk[f, [10, 11]] = pk[f, [11, 10]]
            else:
                lost = (int(f) * 7 + 3) % slots
                keep = [p for p in range(slots) if p != lost]
                pk[f, :slots - 1] = base[f, keep]
                pk[f, slots - 1] = 0
                counts[f] = slots - 1
        d, c = pk.cuda(), counts.cuda()
        for _ in range(30): hp.decode(d, out, packet_counts=c)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(40): hp.decode(d, out, packet_counts=c)
        b.record(); torch.cuda.synchronize()
        res[K] = round(a.elapsed_time(b) / 40 * 1000, 1)
    print(wl, kind, "us per call:", res, " extra per flagged frame (64):", round((res[64] - res[0]) / 64, 2), flush=True)
