#!/usr/bin/env python3
"""alloc_lottery.py arranged for a rocprofv3 --pmc pass: K output sets, L launches into each, in order; prints how
many decode launches precede the sets so that the counter rows can be mapped to the sets by dispatch order."""
import json, os, sys


def channels(O):
    """Post-processing of `LOTTERY_CHANNELS=1 lottery_pmc.sh` (rocprofv3 --output-format json keeps one record per counter
    INSTANCE -- TCC channel x XCC -- where the csv sums them): per output set, per counter, the spread over the instances."""
    import glob, collections
    res = {}
    for d in sorted(glob.glob(O + "/c[0-9]")):
        fs = glob.glob(d + "/**/*results.json", recursive=True)
        if not fs:
            print(d, "no json"); continue
        info = json.loads(open(O + "/cout" + d[-1] + ".txt").read().strip().splitlines()[-1])
        K, L = info["sets"], info["launches_per_set"]
        J = json.load(open(fs[0]))["rocprofiler-sdk-tool"][0]
        names = {c["id"]["handle"]: c["name"] for c in J["counters"]}
        disp = J["callback_records"]["counter_collection"] if "counter_collection" in J.get("callback_records", {}) else J["buffer_records"]["counter_collection"]
        recs = [r for r in disp]
        recs.sort(key=lambda r: r["dispatch_data"]["dispatch_info"]["dispatch_id"])
        recs = recs[-K * L:]
        by = collections.defaultdict(lambda: collections.defaultdict(float))
        for n, r in enumerate(recs):
            seen = collections.Counter()
            for rec in r["records"]:
                h = rec["counter_id"]["handle"] if "counter_id" in rec else rec["id"]["handle"]
                nm = names.get(h, str(h))
                inst = seen[nm]; seen[nm] += 1          # records of one counter in file order = its instances
                by[nm][(n // L, inst)] += float(rec["value"]) / L
        res[d[-2:]] = {"ms": info["ms_per_launch_under_the_profiler"], "kernel": info.get("kernel"), "tile": info.get("tile"), "workload": info.get("workload"), "counters": {}}
        for c, v in by.items():
            n_inst = 1 + max(k[1] for k in v)
            tab = [[round(v[(s_, i_)]) for i_ in range(n_inst)] for s_ in range(K)]
            res[d[-2:]]["counters"][c] = tab
            print(c, "instances", n_inst)
            for s_ in range(K):
                m = sum(tab[s_]) / n_inst
                print("  set", s_, "ms", info["ms_per_launch_under_the_profiler"][s_], "mean", round(m), "min/mean", round(min(tab[s_]) / max(m, 1), 3),
                      "max/mean", round(max(tab[s_]) / max(m, 1), 3))
    json.dump(res, open(O + "/channels.json", "w"))


if len(sys.argv) > 2 and sys.argv[1] == "--channels":
    channels(sys.argv[2])
    sys.exit(0)
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ouster_sdk_amd.device import HotPath
K, L = 8, 6
WL = os.environ.get("LOTTERY_WORKLOAD", "dual")          # round 5: "single" for configs[1]
prof, bits, chan, dst, xyz = bench.WORKLOADS[WL][:5]
H, W, N = bench.H, bench.W, 256
alt, az, shifts, b2l, l2s = bench.synth_calibration()
pool = bench.synth_packets(16, bits=bits, chan=chan)
pk = torch.from_numpy(pool).cuda().repeat(N // 16, 1, 1).contiguous()
hp = HotPath(prof, H, W, 16)
hp.set_pixel_shift_by_row(shifts)
hp.add_lut(b2l, l2s, az, alt)
if os.environ.get("LOTTERY_STREAM"):                      # the persistent kernel with that tile width, else k_decode_wide
    hp.ctx.set_knob("stream", int(os.environ["LOTTERY_STREAM"]))
else:
    hp.ctx.set_knob("wide", int(os.environ.get("LOTTERY_WIDE", "256")))
al = 2 << 20
def slab_set():
    tmpl = hp.alloc_outputs(N, destagger=dst, xyz=xyz)
    names = list(tmpl); sizes = [tmpl[n].numel() * tmpl[n].element_size() for n in names]
    meta = {n: (tmpl[n].dtype, tuple(tmpl[n].shape)) for n in names}
    del tmpl; torch.cuda.empty_cache()
    slab = torch.empty(sum((x + al - 1) // al * al for x in sizes) + al, dtype=torch.uint8, device="cuda")
    off = (-slab.data_ptr()) % al
    out = {}
    for n, nb in zip(names, sizes):
        out[n] = slab[off:off + nb].view(meta[n][0]).view(meta[n][1]); off += (nb + al - 1) // al * al
    return out
sets = [slab_set() for _ in range(K)]
torch.cuda.synchronize()
ms = []
for o in sets:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(L): hp.decode(pk, o)
    b.record(); torch.cuda.synchronize()
    ms.append(round(a.elapsed_time(b) / L, 4))
print(json.dumps({"sets": K, "launches_per_set": L, "ms_per_launch_under_the_profiler": ms, "workload": WL,
                  "kernel": hp.ctx.last_decode_kernel(), "tile": hp.ctx.last_decode_tile()}))
