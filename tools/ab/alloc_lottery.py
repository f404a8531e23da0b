#!/usr/bin/env python3
"""How much of the box-to-box / process-to-process spread is the placement of THIS process's output planes?
One process, one library, one packet buffer: K complete output sets are allocated (all kept alive) and the
decode is timed into each of them in alternating blocks.
  alloc_lottery.py [workload] [wide] [K] [knob:value]     the round-2 form
  alloc_lottery.py stability                               round 6 (profiles/r06_latency/placement_notes.txt item 2): the first set
      timed again and again over 3 s of load, with 64 GB of ballast allocated, after more sets have been allocated and written;
      a second set next to it -- a placement is per allocation and does not change
  alloc_lottery.py slab                                    item 3: per-array allocations against one slab per set (members 2 MB
      aligned, or staggered inside the 2 MB), three rounds"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ouster_sdk_amd.device import HotPath


def _round6(mode):
    import time
    hp, packets, out, *_ = bench._workload_setup("dual", 256)
    prof, bits, chan, dst, xyz = bench.WORKLOADS["dual"][:5]
    hp.ctx.set_knob("tune", 0); hp.ctx.set_knob("stream", 0); hp.ctx.set_knob("wide", 256)

    def clock(o, n=10):
        for _ in range(3): hp.decode(packets, o)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): hp.decode(packets, o)
        b.record(); torch.cuda.synchronize()
        return round(a.elapsed_time(b) / n, 4)
    new_set = lambda: hp.alloc_outputs(256, destagger=dst, xyz=xyz)   # noqa: E731
    if mode == "stability":
        res = {"first_set": [clock(out) for _ in range(5)]}
        t = time.time()
        while time.time() - t < 3.0: clock(out, 50)
        res["after_3s_load"] = [clock(out) for _ in range(3)]
        b = torch.empty(64 << 30, dtype=torch.uint8, device="cuda")
        res["with_64GB_allocated"] = [clock(out) for _ in range(3)]
        del b; torch.cuda.empty_cache()
        o2 = new_set()
        res["second_set"] = [clock(o2) for _ in range(3)]
        o3 = [new_set() for _ in range(6)]
        for o in o3: clock(o)
        res["first_set_after_6_more_sets_written"] = [clock(out) for _ in range(3)]
        res["second_set_again"] = [clock(o2) for _ in range(3)]
    else:
        tmpl = new_set()
        names = list(tmpl); sizes = [tmpl[n].numel() * tmpl[n].element_size() for n in names]
        meta = {n: (tmpl[n].dtype, tuple(tmpl[n].shape)) for n in names}
        del tmpl

        def slab(stagger):
            al = 2 << 20
            big = torch.empty(sum((x + al - 1) // al * al + al for x in sizes) + al, dtype=torch.uint8, device="cuda")
            off, o = (-big.data_ptr()) % al, {}
            for i, (n, nb) in enumerate(zip(names, sizes)):
                st = (i * stagger) % al
                st -= st % 256
                o[n] = big[off + st: off + st + nb].view(meta[n][0]).view(meta[n][1])
                off += (nb + al - 1) // al * al + al
            return o, big
        res, keep = {"first_set": clock(out)}, []
        for _ in range(3):
            o = new_set(); keep.append(o)
            res.setdefault("per_array", []).append(clock(o))
            for S in (0, 4096, 65536 + 4096, 2 * 1024 * 1024 // 17):
                o, big = slab(S); keep.append((o, big))
                res.setdefault(f"slab_stagger_{S}", []).append(clock(o))
            keep.append(torch.empty(6 << 30, dtype=torch.uint8, device="cuda"))
    print(json.dumps(res))


if len(sys.argv) > 1 and sys.argv[1] in ("stability", "slab"):
    _round6(sys.argv[1])
    sys.exit(0)
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
