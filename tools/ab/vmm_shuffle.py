#!/usr/bin/env python3
"""What the allocation lottery is about: PHYSICAL CONTIGUITY.  tools/ab/contig_alloc.py showed that
output planes from hipExtMallocWithFlags(hipDeviceMallocContiguous) decode 45 - 60 % SLOWER than torch's hipMalloc
buffers.  Here the output slab is built with the virtual-memory API from `chunk`-sized physical pieces that are mapped
(a) in the order they were created and (b) in a random permutation -- the same bytes of HBM, a scrambled VA -> PA map.
usage: vmm_shuffle.py <workload> <wide> [draws]     env VMM_CHUNKS_MB (default "2,8,32,128")"""
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


class Raw:
    def __init__(self, ptr, dtype, shape):
        self._p, self.dtype, self.shape = ptr, dtype, shape

    def data_ptr(self):
        return self._p


def vmm_slab(total, chunk, order):
    """order: 'seq' = chunk i at offset i, 'rand' = a random permutation, 'rev' = reversed, 'stride<k>' = interleaved"""
    total = (total + chunk - 1) // chunk * chunk
    n = total // chunk
    va = C.c_void_p()
    ck(hip.hipMemAddressReserve(C.byref(va), C.c_size_t(total), C.c_size_t(0), C.c_void_p(0), C.c_ulonglong(0)), "reserve")
    handles = []
    for _ in range(n):
        h = C.c_void_p()
        ck(hip.hipMemCreate(C.byref(h), C.c_size_t(chunk), C.byref(prop), C.c_ulonglong(0)), "create")
        handles.append(h)
    if order == "seq":
        perm = np.arange(n)
    elif order == "rev":
        perm = np.arange(n)[::-1]
    elif order.startswith("stride"):
        k = int(order[6:])
        perm = np.concatenate([np.arange(i, n, k) for i in range(k)])
    else:
        perm = np.random.default_rng(1234).permutation(n)
    for slot, src in enumerate(perm):
        ck(hip.hipMemMap(C.c_void_p(va.value + slot * chunk), C.c_size_t(chunk), C.c_size_t(0), handles[int(src)],
                         C.c_ulonglong(0)), "map")
    acc = Access(); acc.location.type = 1; acc.location.id = 0; acc.flags = 3
    ck(hip.hipMemSetAccess(va, C.c_size_t(total), C.byref(acc), C.c_size_t(1)), "access")
    return va.value


wl = sys.argv[1] if len(sys.argv) > 1 else "dual"
wide = int(sys.argv[2]) if len(sys.argv) > 2 else 256
K = int(sys.argv[3]) if len(sys.argv) > 3 else 3
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


def t(o, pkb=None):
    pkb = pk if pkb is None else pkb
    for _ in range(3):
        hp.decode(pkb, o)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        hp.decode(pkb, o)
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / 20, 4)


print(json.dumps({"workload": wl, "wide": wide, "granularity_min": gran.value, "torch_tensors_ms": t(tmpl)}), flush=True)
for mb in [int(x) for x in os.environ.get("VMM_CHUNKS_MB", "2,8,32,128").split(",")]:
    ch = max(((mb << 20) + gran.value - 1) // gran.value * gran.value, gran.value)
    row = {}
    for order in os.environ.get("VMM_ORDERS", "seq,rand").split(","):
        try:
            row[order] = [t(carve(vmm_slab(total, ch, order))) for _ in range(K)]
        except Exception as e:
            row[order] = str(e)
    print(json.dumps({f"vmm_{mb}MB_chunks_ms": row}), flush=True)
