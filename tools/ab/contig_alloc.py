#!/usr/bin/env python3
"""Is the allocation lottery about physical contiguity?  K output sets from torch's allocator
(hipMalloc) against K sets from hipExtMallocWithFlags(hipDeviceMallocContiguous), allocated alternately with ballast
in between, the same decode timed into each.
usage: contig_alloc.py <workload> <K> <ballast GB> [mode]   mode: planes (one allocation per plane, default) | slab"""
import ctypes as C, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ouster_sdk_amd.device import HotPath

wl, K, ballast_gb = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "planes"
hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]


class Raw:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def contig(nbytes):
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), nbytes, 0x4)   # hipDeviceMallocContiguous
    if rc != 0:
        raise RuntimeError("hipExtMallocWithFlags(contiguous) failed: %d for %d bytes" % (rc, nbytes))
    return torch.as_tensor(Raw(p.value, nbytes), device="cuda")


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


def like(tmpl, alloc):
    out = {}
    if mode == "slab":
        al = 2 << 20
        sizes = {k: (v.numel() * v.element_size() + al - 1) // al * al for k, v in tmpl.items()}
        slab = alloc(sum(sizes.values()))
        off = 0
        for k, v in tmpl.items():
            out[k] = slab[off:off + v.numel() * v.element_size()].view(v.dtype).view(v.shape)
            off += sizes[k]
        return out
    for k, v in tmpl.items():
        out[k] = alloc(v.numel() * v.element_size()).view(v.dtype).view(v.shape)
    return out


tmpl = hp.alloc_outputs(N, destagger=dst, xyz=xyz)
sets, kinds, held = [], [], []
for i in range(K):
    for kind, alloc in (("torch", lambda n: torch.empty(n, dtype=torch.uint8, device="cuda")), ("contig", contig)):
        sets.append(like(tmpl, alloc))
        kinds.append(kind)
        if ballast_gb > 0:
            held.append(torch.empty(int(ballast_gb * (1 << 30)), dtype=torch.uint8, device="cuda"))
del tmpl


def t(o):
    for _ in range(2):
        hp.decode(pk, o)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(12):
        hp.decode(pk, o)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 12


res = {"workload": wl, "mode": mode, "ballast_gb": ballast_gb, "torch": [], "contig": []}
for o, kind in zip(sets, kinds):
    res[kind].append(round(float(np.median([t(o) for _ in range(3)])), 4))
print(json.dumps(res))
