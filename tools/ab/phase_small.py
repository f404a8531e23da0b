#!/usr/bin/env python3
"""Phase timeline of the small-batch decode (instrumented library, see phase_timing.sh): s_memtime ticks (100 MHz)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
buf = torch.zeros((1 << 14, 16), dtype=torch.int64, device="cuda")
os.environ["OUSTER_HIP_PHASE_BUF"] = hex(buf.data_ptr())
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
knobs = dict(kv.split("=") for kv in sys.argv[2:])
hp, packets, out, *_ = bench._workload_setup("fused4" if n == 4 else "dual", n)
for k, v in knobs.items():
    hp.ctx.set_knob(k, int(v))
for _ in range(10):
    hp.decode(packets, out)
torch.cuda.synchronize()
buf.zero_()
torch.cuda.synchronize()
hp.decode(packets, out)
torch.cuda.synchronize()
t = buf.cpu().numpy().astype(np.float64)
t = t[t[:, 0] != 0]
print(hp.ctx.last_decode_kernel(), hp.ctx.last_decode_tile(), "workgroups", len(t))
T = 100.0  # ticks per us
names = {"resolve: header loads + init": (8, 9), "resolve A": (9, 10), "resolve B (serial)": (10, 11), "resolve C": (11, 12), "resolve final": (12, 13),
         "lead / gap -> tile start": (13, 0), "tile prologue": (0, 1), "staging": (1, 2), "classify": (2, 3), "rows": (3, 4), "store drain": (4, 5)}
for nme, (a, b) in names.items():
    ok = (t[:, a] != 0) & (t[:, b] != 0)
    if ok.any():
        d = (t[ok, b] - t[ok, a]) / T
        print(f"{nme:32s} median {np.median(d):7.2f} us   p90 {np.percentile(d, 90):7.2f}   max {d.max():7.2f}")
first = t[:, 8][t[:, 8] != 0]
if len(first):
    ok = t[:, 8] != 0
    print("workgroup life (resolve start -> stores drained): median %.2f us" % np.median((t[ok, 5] - t[ok, 8]) / T))
else:
    print("workgroup life (tile start -> stores drained): median %.2f us" % np.median((t[:, 5] - t[:, 0]) / T))
