"""The kernel-variant tuner's verdicts persist (include/ouster_hip.h, ouster_hip_ctx_set_tuning_cache): a process that finds
its workload in the cache file launches the recorded variant from its FIRST decode and never times a candidate; the planes and
clouds it writes are those of the process that measured, bit for bit."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r'''
import hashlib, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from ouster_sdk_amd.device import HotPath
H, W, CPP = bench.H, bench.W, bench.CPP
alt, az, shifts, b2l, l2s = bench.synth_calibration()
hp = HotPath(bench.PROFILE, H, W, CPP)
if os.environ.get("CACHE"):
    hp.ctx.set_tuning_cache(os.environ["CACHE"])
hp.set_pixel_shift_by_row(shifts)
hp.add_lut(b2l, l2s, az, alt)
F = 128
pk = torch.from_numpy(bench.synth_packets(8)).cuda().repeat(F // 8, 1, 1).contiguous()
out = hp.alloc_outputs(F, destagger=["RANGE", "RANGE2", "REFLECTIVITY", "REFLECTIVITY2"], xyz=["RANGE", "RANGE2"])
torch.empty(1, device="cuda").zero_(); torch.cuda.synchronize()     # the process's first kernel launch is not the decode's
tuners, kernels, ms = [], [], []
for i in range(48):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); hp.decode(pk, out); b.record(); torch.cuda.synchronize()
    ms.append(a.elapsed_time(b)); tuners.append(hp.ctx.last_decode_tuner()); kernels.append("%s %dx%d" % ((hp.ctx.last_decode_kernel(),) + hp.ctx.last_decode_tile()))
h = hashlib.sha256()
for k in sorted(out):
    h.update(out[k].cpu().numpy().tobytes())
print(json.dumps({"tuners": tuners, "kernels": kernels, "ms": ms, "sha": h.hexdigest()}))
'''


def _run(cache):
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("OUSTER_HIP_TUNING_CACHE", None)
    if cache:
        env["CACHE"] = cache
    p = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_second_process_launches_the_cached_variant_from_its_first_call(tmp_path):
    cache = str(tmp_path / "tuning.cache")
    a = _run(cache)
    assert a["tuners"][0] == "measuring" and a["tuners"][-1] == "measured", a["tuners"]
    lines = [ln.split() for ln in open(cache).read().splitlines()]
    assert lines and all(ln[0] == "v1" and len(ln) == 5 for ln in lines), lines
    b = _run(cache)
    assert set(b["tuners"]) == {"cache"}, b["tuners"]                      # not one timing call
    assert b["kernels"][0] == a["kernels"][-1] and len(set(b["kernels"])) == 1, (a["kernels"][-1], b["kernels"][:3])
    assert a["sha"] == b["sha"]                                            # same planes, headers and clouds, bit for bit
    steady = sorted(b["ms"][24:])[len(b["ms"][24:]) // 2]
    early = sorted(b["ms"][1:6])[2]                                        # (call 0 loads the kernel's code object)
    assert early <= 1.10 * steady, (early, steady, b["ms"][:8])
    # a process without the cache measures again; a cache written for another device / library version is ignored
    with open(cache, "w") as f:
        for ln in lines:
            f.write(" ".join([ln[0], "some_other_gpu:1:x:v0"] + ln[2:]) + "\n")
    c = _run(cache)
    assert c["tuners"][0] == "measuring" and c["sha"] == a["sha"]
    print("first calls with the cache:", [round(x, 3) for x in b["ms"][:6]], "steady", round(steady, 3), "kernel", b["kernels"][0],
          "| measuring process:", [round(x, 3) for x in a["ms"][:6]])
