#!/usr/bin/env python3
"""Same-process A/B of decode-kernel variants selected by knobs: every variant decodes the SAME packet buffer into
the SAME output tensors (identical physical placement) in alternating blocks of 20 calls; outputs are compared byte
for byte with the first variant's.
usage: stream_ab.py <workload> [frames] name=knob:value[,knob:value...] ...
  e.g. stream_ab.py dual 256 wide=stream:0,wide:256 s256=stream:256 s256nw=stream:256,stream_wait:0"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from ouster_sdk_amd.device import HotPath

wl = sys.argv[1]
args = sys.argv[2:]
N = 256
if args and args[0].isdigit():
    N = int(args.pop(0))
variants = []
for a in args:
    name, spec = a.split("=", 1)
    variants.append((name, [kv.split(":") for kv in spec.split(",") if kv]))
prof, bits, chan, dst, xyz = bench.WORKLOADS[wl][:5]
H, W = bench.H, bench.W
alt, az, shifts, b2l, l2s = bench.synth_calibration()
pool = bench.synth_packets(16, bits=bits, chan=chan)
pk = torch.from_numpy(pool).cuda().repeat(N // 16, 1, 1).contiguous()
hps, out = {}, None
for name, knobs in variants:
    hp = HotPath(prof, H, W, 16)
    hp.set_pixel_shift_by_row(shifts)
    hp.add_lut(b2l, l2s, az, alt)
    hp.ctx.set_knob("tune", 0)
    for k, v in knobs:
        hp.ctx.set_knob(k, int(v))
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
    for name, knobs in variants:
        hps[name].ctx.set_knob("tune", 0)
ref = None
times = {n: [] for n in hps}
kern = {}
for rnd in range(int(os.environ.get("AB_ROUNDS", "5"))):
    for name, hp in hps.items():
        for t in out.values():
            t.view(torch.uint8).fill_(0x3C)
        hp.decode(pk, out)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        hp.ctx.timing(True)
        a.record()
        for _ in range(20):
            hp.decode(pk, out)
        b.record()
        torch.cuda.synchronize()
        kern.setdefault(name, []).append(hp.ctx.timing_read()[0])
        hp.ctx.timing(False)
        times[name].append(a.elapsed_time(b) / 20)
        if ref is None:
            ref = {k: v.clone() for k, v in out.items()}
        bad = [k for k, v in ref.items() if not torch.equal(v.view(torch.uint8), out[k].view(torch.uint8))]
        assert not bad, (name, bad)
abytes = bench.algorithmic_bytes_per_frame(wl) * N
print(json.dumps({"workload": wl, "frames": N,
                  "kernels": {n: [h.ctx.last_decode_kernel()] + list(h.ctx.last_decode_tile()) for n, h in hps.items()},
                  "ms_per_call_median": {n: round(float(np.median(t)), 4) for n, t in times.items()},
                  "ms_per_call_min": {n: round(float(np.min(t)), 4) for n, t in times.items()},
                  "kernel_ms_median": {n: round(float(np.median(t)), 4) for n, t in kern.items()},
                  "frac_of_8TBps_median_call": {n: round(abytes / (float(np.median(t)) * 1e-3) / 8e12, 4) for n, t in times.items()}}))
