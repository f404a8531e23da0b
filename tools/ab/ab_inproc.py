#!/usr/bin/env python3
"""Same-process A/B of library builds: every variant decodes the SAME packet buffer into the SAME output
tensors (identical physical placement -- separate processes on one box differ by +-6 % for that reason alone),
in alternating blocks.  usage: ab_inproc.py <workload> <wide knob> name=path.so [name=path.so ...]
(name 'base' with an empty path = the product build; name=@knob:value = the product build with that knob set)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ouster_sdk_amd import _capi
from ouster_sdk_amd.device import HotPath

wl, wide = sys.argv[1], int(sys.argv[2])
os.environ["OUSTER_HIP_WIDE"] = str(wide)
variants = [a.split("=", 1) for a in sys.argv[3:]]
prof, bits, chan, dst, xyz = bench.WORKLOADS[wl][:5]
H, W, N = bench.H, bench.W, 256
alt, az, shifts, b2l, l2s = bench.synth_calibration()
pool = bench.synth_packets(16, bits=bits, chan=chan)
pk = torch.from_numpy(pool).cuda().repeat(N // 16, 1, 1).contiguous()
hps, out = {}, None
for name, path in variants:
    knob = path[1:].split(":") if path.startswith("@") else None
    lib = _capi.load_hip(os.path.join(ROOT, path)) if path and not knob else None
    hp = HotPath(prof, H, W, 16, lib=lib)

    hp.set_pixel_shift_by_row(shifts)
    hp.add_lut(b2l, l2s, az, alt)
    try:
        hp.ctx.set_knob("wide", wide)
    except Exception:          # the round-1 library has no knobs: it reads OUSTER_HIP_WIDE from the environment
        pass
    if knob:
        hp.ctx.set_knob(knob[0], int(knob[1]))
    if out is None:
        out = hp.alloc_outputs(N, destagger=dst, xyz=xyz)
    hps[name] = hp
    for _ in range(3):
        hp.decode(pk, out)
torch.cuda.synchronize()
if os.environ.get("AB_PLACEMENT"):     # first find a fast place for the buffers, then compare there
    first = hps[variants[0][0]]
    pk, out, rep = first.pick_placement(pk, lambda: first.alloc_outputs(N, destagger=dst, xyz=xyz),
                                        tries=int(os.environ["AB_PLACEMENT"]), stride_gb=4.0)
    print("placement:", min(rep["output_sets_ms"]), max(rep["output_sets_ms"]), file=sys.stderr)
    for hp in hps.values():
        try:
            hp.ctx.set_knob("wide", wide)
        except Exception:
            pass
        for _ in range(3):
            hp.decode(pk, out)
    torch.cuda.synchronize()
ref = {k: v.clone() for k, v in out.items() if k != "frame_meta"}
times = {n: [] for n in hps}
for rnd in range(6):
    for name, hp in hps.items():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        hp.decode(pk, out)
        a.record()
        for _ in range(20):
            hp.decode(pk, out)
        b.record()
        torch.cuda.synchronize()
        times[name].append(a.elapsed_time(b) / 20)
        for k, v in ref.items():
            if k in ("frame_meta", "gate_counts") or os.environ.get("AB_NOCHECK"): continue   # ablation builds differ by design
            assert torch.equal(v.view(torch.uint8), out[k].view(torch.uint8)), (name, k)
print(json.dumps({"workload": wl, "wide": wide, "tiles": {n: list(h.ctx.last_decode_tile()) for n, h in hps.items()},
                  "ms_per_call_median": {n: round(float(np.median(t)), 4) for n, t in times.items()},
                  "ms_per_call_min": {n: round(float(np.min(t)), 4) for n, t in times.items()}}))
