#!/usr/bin/env python3
"""Small-batch decode under different kernel choices (knobs): us per call, pipelined and kernel-only.
usage: python tools/ab/small_batch.py [frames ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

CONFIGS = [
    ("default", {}),
    ("small=0 (narrow + fixup)", {"small": 0}),
    ("small=0 fixup=0 (narrow only)", {"small": 0, "fixup": 0}),
    ("wide256 rows4 optimistic + fixup", {"small": 0, "wide": 256, "wide_rows": 4, "wide_min_blocks": 0, "stream": 0}),
    ("wide256 rows8 optimistic + fixup", {"small": 0, "wide": 256, "wide_rows": 8, "wide_min_blocks": 0, "stream": 0}),
    ("wide256 rows8 optimistic, fixup=0", {"small": 0, "wide": 256, "wide_rows": 8, "wide_min_blocks": 0, "stream": 0, "fixup": 0}),
    ("wide256 rows16 optimistic, fixup=0", {"small": 0, "wide": 256, "wide_rows": 16, "wide_min_blocks": 0, "stream": 0, "fixup": 0}),
    ("wide128 rows8 optimistic, fixup=0", {"small": 0, "wide": 128, "wide_rows": 8, "wide_min_blocks": 0, "stream": 0, "fixup": 0}),
    ("wide256 rows32 optimistic + fixup", {"small": 0, "wide": 256, "wide_rows": 32, "wide_min_blocks": 0, "stream": 0}),
    ("wide256 rows16 optimistic + fixup", {"small": 0, "wide": 256, "wide_rows": 16, "wide_min_blocks": 0, "stream": 0}),
    ("wide128 rows16 optimistic + fixup", {"small": 0, "wide": 128, "wide_rows": 16, "wide_min_blocks": 0, "stream": 0}),
    ("one launch (small=2)", {"small": 2}),
]

for n in [int(x) for x in sys.argv[1:]] or [1, 4]:
    for label, knobs in CONFIGS:
        hp, packets, out, profile, shifts, lut_args, n_ret, _ = bench._workload_setup("fused4" if n == 4 else "dual", n)
        for k, v in knobs.items():
            hp.ctx.set_knob(k, v)
        inputs = [packets, packets.clone()]
        for _ in range(24):
            hp.decode(packets, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(300):
            hp.decode(inputs[i & 1], out)
        torch.cuda.synchronize()
        pipe = (time.perf_counter() - t0) / 300
        hp.ctx.timing(True)
        for i in range(30):
            hp.decode(inputs[i & 1], out)
        torch.cuda.synchronize()
        kms, _ = hp.ctx.timing_read()
        hp.ctx.timing(False)
        tc, tr = hp.ctx.last_decode_tile()
        print(f"{n} frames  {label:42s} {hp.ctx.last_decode_kernel():24s} {tc}x{tr:<4d} kernel {kms*1e3:7.2f} us  pipelined {pipe*1e6:7.2f} us", flush=True)
        del hp, packets, out, inputs
