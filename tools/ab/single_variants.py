#!/usr/bin/env python3
"""configs[1] (12 B/px) under every kernel variant, same process and buffers: ms per call and fraction of 8 TB/s."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "single"
N = 256
hp, packets, out, *_ = bench._workload_setup(wl, N, pool_frames=8)
inputs = [packets, packets.clone()]
nbytes = bench.algorithmic_bytes_per_frame(wl) * N
VARIANTS = [("auto", {}), ("stream128", {"stream": 128}), ("stream256", {"stream": 256}), ("wide128", {"stream": 0, "wide": 128}),
            ("wide256", {"stream": 0, "wide": 256}), ("wide64", {"stream": 0, "wide": 64}), ("narrow", {"stream": 0, "wide": 0}),
            ("stream128 loader0", {"stream": 128, "stream_loader": 0}), ("stream128 rows16", {"stream": 128, "stream_rows": 16}),
            ("stream128 rows64", {"stream": 128, "stream_rows": 64}), ("wide128 rows16", {"stream": 0, "wide": 128, "wide_rows": 16}),
            ("wide128 rows64", {"stream": 0, "wide": 128, "wide_rows": 64})]
DEFAULTS = {"stream": -1, "wide": -1, "stream_loader": 4, "stream_rows": 0, "wide_rows": 0}
for rep in range(2):
    for name, knobs in VARIANTS:
        for k, v in DEFAULTS.items():
            hp.ctx.set_knob(k, v)
        for k, v in knobs.items():
            hp.ctx.set_knob(k, v)
        hp.ctx.set_knob("retune", 1)
        for _ in range(20):
            hp.decode(packets, out)
        torch.cuda.synchronize()
        hp.ctx.timing(True)
        for i in range(20):
            hp.decode(inputs[i & 1], out)
        torch.cuda.synchronize()
        kms, _ = hp.ctx.timing_read()
        hp.ctx.timing(False)
        tc, tr = hp.ctx.last_decode_tile()
        print(f"{wl} {name:22s} {hp.ctx.last_decode_kernel():18s} {tc}x{tr:<4d} {kms:.4f} ms  {nbytes / (kms * 1e-3) / 8e12:.4f}", flush=True)
