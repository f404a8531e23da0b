"""Which destaggered outputs make k_decode layout-sensitive?  Same process, two layouts."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ouster_sdk_amd.device import HotPath

H, W, F = 128, 2048, 256
alt, az, shifts, b2l, l2s = bench.synth_calibration()
hp = HotPath("RNG15_RFL8_NIR8_DUAL", H, W, 16)
hp.add_lut(b2l, l2s, az, alt)
pool = bench.synth_packets(16, seed=1)
packets = torch.from_numpy(pool).cuda().repeat(F // 16, 1, 1).contiguous()
xyz = ["RANGE", "RANGE2"]


def run(out, steps=20):
    for _ in range(4):
        hp.decode(packets, out)
    torch.cuda.synchronize()
    hp.ctx.timing(True)
    for _ in range(steps):
        hp.decode(packets, out)
    torch.cuda.synchronize()
    ms, n = hp.ctx.timing_read()
    hp.ctx.timing(False)
    return round(ms, 4)


def slab_like(tmpl_items, sizes):
    al = lambda n: (n + (1 << 21) - 1) & ~((1 << 21) - 1)
    slab = torch.empty(sum(al(n) for n in sizes.values()) + (1400 << 20), dtype=torch.uint8, device="cuda")
    o = (-slab.data_ptr()) % (1 << 21)
    out = {}
    for k, t in tmpl_items:
        n = sizes[k]
        out[k] = slab[o:o + n].view(t.dtype).view((F,) + tuple(t.shape[1:]))
        o += al(n)
    return out, slab


res = {}
for shname, sh in (("real_shifts", shifts), ("zero_shifts", np.zeros_like(shifts)), ("shift_32cols", np.full_like(shifts, 32))):
    hp.set_pixel_shift_by_row(sh)
    for name, dst in (("all4", ["RANGE", "RANGE2", "REFLECTIVITY", "REFLECTIVITY2"]), ("u32_only", ["RANGE", "RANGE2"]),
                      ("u8_only", ["REFLECTIVITY", "REFLECTIVITY2"]), ("none", [])):
        if shname != "real_shifts" and name not in ("all4",):
            continue
        sep = hp.alloc_outputs(F, destagger=dst, xyz=xyz)
        t_sep = run(sep)
        sizes = {k: v.numel() * v.element_size() for k, v in sep.items()}
        tmpl = list(hp.alloc_outputs(1, destagger=dst, xyz=xyz).items())
        del sep
        torch.cuda.empty_cache()
        out, slab = slab_like(tmpl, sizes)
        t_slab = run(out)
        del out, slab
        torch.cuda.empty_cache()
        res[f"{shname}/{name}"] = {"separate": t_sep, "slab": t_slab}
        print(shname, name, res[f"{shname}/{name}"], flush=True)
print(json.dumps(res))
