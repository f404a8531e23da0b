#!/usr/bin/env python3
"""One flagged frame of several kinds, interleaved in one process (same buffers): us per call over the clean batch."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
N = 256
hp, base, out, *_ = bench._workload_setup("dual", N, pool_frames=8)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    hp.ctx.set_knob(k, int(v))
base = base.cpu()
slots = base.shape[1]

def make(kind, f=0):
    pk = base.clone()
    counts = torch.full((N,), slots, dtype=torch.int32)
    if kind == "swap":
        pk[f, [10, 11]] = pk[f, [11, 10]]
    elif kind == "swapshort":
        pk[f, [10, 11]] = pk[f, [11, 10]]; pk[f, slots - 1] = 0; counts[f] = slots - 1
    elif kind == "short":          # only the count differs: not flagged at all
        pk[f, slots - 1] = 0; counts[f] = slots - 1
    elif kind == "rot":
        pk[f, 20:30] = base[f, [29] + list(range(20, 29))]
    elif kind.startswith("compact"):
        lost = int(kind[7:] or 3)
        keep = [p for p in range(slots) if p != lost]
        pk[f, :slots - 1] = base[f, keep]; pk[f, slots - 1] = 0; counts[f] = slots - 1
    elif kind.startswith("shift"):  # like compact, but the count stays: the last slot holds a copy of the last packet
        lost = int(kind[5:] or 3)
        keep = [p for p in range(slots) if p != lost]
        pk[f, :slots - 1] = base[f, keep]; pk[f, slots - 1] = base[f, slots - 1]
    return pk.cuda(), counts.cuda()

def clock(d, c):
    for _ in range(25): hp.decode(d, out, packet_counts=c)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(40): hp.decode(d, out, packet_counts=c)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 40 * 1000

kinds = ["clean", "swap", "short", "swapshort", "rot", "compact3", "compact100", "compact124", "shift3", "shift100"]
bufs = {k: make(k) for k in kinds}
for rep in range(3):
    row = {k: round(clock(*bufs[k]), 1) for k in kinds}
    print({k: round(v - row["clean"], 1) for k, v in row.items()}, "clean", row["clean"], flush=True)

# the same buffers through the general mapping of a second context (k_slotmap + k_decode_wide from the maps): every byte equal?
hp2, _, out2, *_ = bench._workload_setup("dual", N, pool_frames=8)
hp2.ctx.set_knob("fast", 0)
for k in kinds:
    d, c = bufs[k]
    for t in out.values(): t.view(torch.uint8).fill_(0xCD)
    for t in out2.values(): t.view(torch.uint8).fill_(0xEE)
    hp.decode(d, out, packet_counts=c); hp2.decode(d, out2, packet_counts=c)
    torch.cuda.synchronize()
    bad = [n for n in out if n != "frame_meta" and not torch.equal(out[n].view(torch.uint8), out2[n].view(torch.uint8))]
    print("check", k, "equal" if not bad else ("DIFFERENT: " + ",".join(bad)), hp2.ctx.last_decode_kernel(), flush=True)
