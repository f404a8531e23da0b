#!/usr/bin/env python3
"""ab_inproc.py for any profile: the packets come from the oracle's synthesizer (a tool may use it; the product and
bench.py do not).  usage: ab_profile.py <profile> <W> <wide knob> name=path.so|name=@knob:value ..."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from ouster_sdk_amd import _capi
from ouster_sdk_amd.device import HotPath
profile, W, wide = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
variants = [a.split("=", 1) for a in sys.argv[4:]]
H, N = 128, 256
cal = O.synthetic_calib(h=H, w=W, profile=profile)
packets, _ = O.synth_packets(cal, 8, with_window=True)
pk = torch.from_numpy(packets).cuda().repeat(N // 8, 1, 1).contiguous()
hps, out = {}, None
names = None
for name, path in variants:
    knob = path[1:].split(":") if path.startswith("@") else None
    lib = _capi.load_hip(os.path.join(ROOT, path)) if path and not knob else None
    hp = HotPath(profile, H, W, cal.cpp, header_type=cal.header_type, lib=lib)
    hp.set_pixel_shift_by_row(cal.pixel_shift_by_row)
    hp.add_lut(cal.beam_to_lidar, cal.lut_transform(True), cal.beam_azimuth_angles, cal.beam_altitude_angles)
    hp.ctx.set_knob("wide", wide)
    if knob:
        hp.ctx.set_knob(knob[0], int(knob[1]))
    if out is None:
        fn = [n for n, _ in hp.fields]
        dst = [n for n in ("RANGE", "RANGE2", "REFLECTIVITY", "REFLECTIVITY2") if n in fn]
        xyz = [n for n in ("RANGE", "RANGE2") if n in fn]
        make = lambda: hp0.alloc_outputs(N, destagger=dst, xyz=xyz)  # noqa: E731
        hp0 = hp
        out = make()
    hps[name] = hp
    for _ in range(3):
        hp.decode(pk, out)
torch.cuda.synchronize()
if os.environ.get("AB_PLACEMENT"):
    pk, out, rep = hp0.pick_placement(pk, make, tries=int(os.environ["AB_PLACEMENT"]), stride_gb=4.0)
    for hp in hps.values():
        hp.ctx.set_knob("wide", wide)
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
            assert torch.equal(v.view(torch.uint8), out[k].view(torch.uint8)), (name, k)
print(json.dumps({"profile": profile, "W": W, "tiles": {n: list(h.ctx.last_decode_tile()) for n, h in hps.items()},
                  "ms_per_call_median": {n: round(float(np.median(t)), 4) for n, t in times.items()}}))
