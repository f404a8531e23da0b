#!/usr/bin/env python3
"""Can the allocation lottery (alloc_lottery.py) be avoided by HOW the memory is obtained?  Output slabs from
hipMalloc (what torch and DeviceBuffer use) against slabs built with the virtual-memory API (hipMemCreate +
hipMemMap) from physical chunks of a chosen size; the same decode into each, one process."""
import ctypes as C, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ouster_sdk_amd.device import HotPath

torch.zeros(1, device="cuda")
path = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]
hip = C.CDLL(path)

class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]
class Flags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]
class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", Loc),
                ("win32HandleMetaData", C.c_void_p), ("allocFlags", Flags)]
class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]

def ck(e, what):
    if e != 0:
        raise RuntimeError(f"{what}: hip error {e}")

prop = Prop(); prop.type = 1; prop.location.type = 1; prop.location.id = 0
gran = C.c_size_t()
ck(hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), C.c_int(0)), "granularity(min)")
gran_rec = C.c_size_t()
ck(hip.hipMemGetAllocationGranularity(C.byref(gran_rec), C.byref(prop), C.c_int(1)), "granularity(rec)")

class Raw:
    """what HotPath.decode needs of a tensor"""
    def __init__(self, ptr, dtype, shape):
        self._p, self.dtype, self.shape = ptr, dtype, shape
    def data_ptr(self): return self._p

def vmm_slab(total, chunk):
    total = (total + chunk - 1) // chunk * chunk
    va = C.c_void_p()
    ck(hip.hipMemAddressReserve(C.byref(va), C.c_size_t(total), C.c_size_t(0), C.c_void_p(0), C.c_ulonglong(0)), "reserve")
    handles = []
    for off in range(0, total, chunk):
        h = C.c_void_p()
        ck(hip.hipMemCreate(C.byref(h), C.c_size_t(chunk), C.byref(prop), C.c_ulonglong(0)), "create")
        ck(hip.hipMemMap(C.c_void_p(va.value + off), C.c_size_t(chunk), C.c_size_t(0), h, C.c_ulonglong(0)), "map")
        handles.append(h)
    acc = Access(); acc.location.type = 1; acc.location.id = 0; acc.flags = 3
    ck(hip.hipMemSetAccess(va, C.c_size_t(total), C.byref(acc), C.c_size_t(1)), "access")
    return va.value, total, handles

def malloc_slab(total):
    p = C.c_void_p()
    ck(hip.hipMalloc(C.byref(p), C.c_size_t(total)), "hipMalloc")
    return p.value

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
sizes = [tmpl[n].numel() * tmpl[n].element_size() for n in names]
total = sum((s + MB2 - 1) // MB2 * MB2 for s in sizes)

def carve(base):
    out, off = {}, 0
    for n, nb in zip(names, sizes):
        out[n] = Raw(base + off, tmpl[n].dtype, tmpl[n].shape)
        off += (nb + MB2 - 1) // MB2 * MB2
    return out

def t(o):
    for _ in range(3): hp.decode(pk, o)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): hp.decode(pk, o)
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / 20, 4)

for _ in range(14): hp.decode(pk, tmpl)
res = {"workload": wl, "wide": wide, "granularity_min": gran.value, "granularity_recommended": gran_rec.value,
       "torch_tensors_ms": t(tmpl)}
print(json.dumps(res), flush=True)
print(json.dumps({"hipMalloc_slab_ms": [t(carve(malloc_slab(total))) for _ in range(int(os.environ.get("VMM_DRAWS", "6")))]}), flush=True)
K = int(os.environ.get("VMM_DRAWS", "6"))
for mb in [int(x) for x in os.environ.get("VMM_CHUNKS_MB", "16,32,64,128,256").split(",")]:
    ch = ((mb << 20) + gran.value - 1) // gran.value * gran.value
    try:
        print(json.dumps({f"vmm_{mb}MB_chunks_ms": [t(carve(vmm_slab(total, ch)[0])) for _ in range(K)]}), flush=True)
    except Exception as e:
        print(json.dumps({f"vmm_{mb}MB_chunks_ms": str(e)}), flush=True)
