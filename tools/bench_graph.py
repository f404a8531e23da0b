#!/usr/bin/env python3
"""Single-frame latency of the batched C ABI: eager launches vs one HIP-graph replay."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ouster_sdk_amd.device import HotPath

H, W = 128, 2048
res = {}
for F in (1, 4):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        alt, az, shifts, b2l, l2s = bench.synth_calibration()
        hp = HotPath("RNG15_RFL8_NIR8_DUAL", H, W, 16)
        hp.set_pixel_shift_by_row(shifts)
        hp.add_lut(b2l, l2s, az, alt)
        pk = torch.from_numpy(bench.synth_packets(F, seed=3)).cuda()
        out = hp.alloc_outputs(F, destagger=["RANGE", "RANGE2", "REFLECTIVITY", "REFLECTIVITY2"],
                               xyz=["RANGE", "RANGE2"])
        for _ in range(5):
            hp.decode(pk, out)
        s.synchronize()
        n = 2000
        t0 = time.perf_counter()
        for _ in range(n):
            hp.decode(pk, out)
        s.synchronize()
        eager = (time.perf_counter() - t0) / n * 1e6
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            hp.decode(pk, out)
        for _ in range(5):
            g.replay()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        s.synchronize()
        graph = (time.perf_counter() - t0) / n * 1e6
        res[f"{F}_frames"] = {"eager_us_per_call": round(eager, 2), "graph_us_per_replay": round(graph, 2)}
print(json.dumps(res))
