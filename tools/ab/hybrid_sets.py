#!/usr/bin/env python3
"""Is a slow / fast draw of the allocation lottery the sum of its buffers' own draws?  K output sets
(one allocation per plane) with ballast in between; the fastest (A) and the slowest (B) are mixed buffer group by
buffer group and the full decode is timed into every hybrid.
usage: hybrid_sets.py <workload> <K> <ballast GB>"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ouster_sdk_amd.device import HotPath

wl, K, ballast_gb = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
prof, bits, chan, dst, xyz = bench.WORKLOADS[wl][:5]
H, W, N = bench.H, bench.W, 256
alt, az, shifts, b2l, l2s = bench.synth_calibration()
pool = bench.synth_packets(16, bits=bits, chan=chan)
pk = torch.from_numpy(pool).cuda().repeat(N // 16, 1, 1).contiguous()
hp = HotPath(prof, H, W, 16)
hp.set_pixel_shift_by_row(shifts)
hp.add_lut(b2l, l2s, az, alt)
hp.ctx.set_knob("tune", 0)
hp.ctx.set_knob("stream", 0)
hp.ctx.set_knob("wide", 256)
sets, held = [], []
for _ in range(K):
    sets.append(hp.alloc_outputs(N, destagger=dst, xyz=xyz))
    if ballast_gb > 0:
        held.append(torch.empty(int(ballast_gb * (1 << 30)), dtype=torch.uint8, device="cuda"))


def t(o):
    for _ in range(2):
        hp.decode(pk, o)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(16):
        hp.decode(pk, o)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 16


def med(o):
    return round(float(np.median([t(o) for _ in range(3)])), 4)


ms = [med(o) for o in sets]
A, B = sets[int(np.argmin(ms))], sets[int(np.argmax(ms))]
groups = {
    "xyz": [k for k in A if k.startswith("xyz:")],
    "u32_planes": [k for k in A if not k.startswith(("xyz:", "destaggered:")) and A[k].element_size() == 4 and A[k].dim() == 3],
    "u8_u16_planes": [k for k in A if not k.startswith(("xyz:", "destaggered:")) and A[k].element_size() < 4 and A[k].dim() == 3],
    "destaggered": [k for k in A if k.startswith("destaggered:")],
    "headers": [k for k in A if A[k].dim() != 3 and not k.startswith("xyz:")],
}
res = {"workload": wl, "sets_ms": ms, "A_ms": min(ms), "B_ms": max(ms), "A_with_group_from_B": {}, "B_with_group_from_A": {},
       "A_with_single_buffer_from_B": {}}
for g, keys in groups.items():
    h = dict(A); h.update({k: B[k] for k in keys})
    res["A_with_group_from_B"][g] = med(h)
    h = dict(B); h.update({k: A[k] for k in keys})
    res["B_with_group_from_A"][g] = med(h)
for k in A:
    if A[k].dim() == 3 or k.startswith("xyz:"):
        h = dict(A); h[k] = B[k]
        res["A_with_single_buffer_from_B"][k] = med(h)
res["A_again"], res["B_again"] = med(A), med(B)
# the packet buffer's own draw
pk2 = [pk.clone() for _ in range(3)]
res["A_with_packet_copies"] = []
for p2 in pk2:
    pk_save = pk
    pk = p2
    res["A_with_packet_copies"].append(med(A))
    pk = pk_save
print(json.dumps(res))
