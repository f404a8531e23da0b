#!/usr/bin/env python3
"""Phase timeline of k_decode_wide workgroups (instrumented library, see phase_timing.sh)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
wl = sys.argv[1] if len(sys.argv) > 1 else "single"
NWG = 1 << 16
buf = torch.zeros((NWG, 16), dtype=torch.int64, device="cuda")
os.environ["OUSTER_HIP_PHASE_BUF"] = hex(buf.data_ptr())
os.environ.setdefault("OUSTER_HIP_WIDE", "128")
import bench
from ouster_sdk_amd.device import HotPath
N = 256
prof, bits, chan, dst, xyz = bench.WORKLOADS[wl][:5]
H, W = bench.H, bench.W
sys.argv = ["bench.py", "--workload", wl, "--steps", "20", "--warmup", "3", "--no-cpu"]
import io, contextlib
cap = io.StringIO()
with contextlib.redirect_stdout(cap):
    bench.main()                      # the usual bench line (kernel time for the calibration) ...
torch.cuda.synchronize()
line = json.loads([l for l in cap.getvalue().splitlines() if l.startswith("{")][-1])
kern_us = line["roofline"]["kernel_ms_avg"] * 1e3
print("kernel", line["roofline"]["kernel"], kern_us, "us")
# ... then one more launch of the same shape into a clean stamp buffer
alt, az, shifts, b2l, l2s = bench.synth_calibration()
hp = HotPath(prof, H, W, 16)
hp.set_pixel_shift_by_row(shifts)
lut = hp.add_lut(b2l, l2s, az, alt)
pool = bench.synth_packets(16, bits=bits, chan=chan)
pk = torch.from_numpy(pool).cuda().repeat(N // 16, 1, 1).contiguous()
out = hp.alloc_outputs(N, destagger=dst, xyz=xyz)
for _ in range(3):
    hp.decode(pk, out)
torch.cuda.synchronize()
buf.zero_()
torch.cuda.synchronize()
hp.decode(pk, out)
torch.cuda.synchronize()
t = buf.cpu().numpy()
t = t[t[:, 0] != 0]
spans = [float(t[t[:, 7] == k][:, 5].max() - t[t[:, 7] == k][:, 0].min()) for k in np.unique(t[:, 7])]
print("workgroups of the last launch:", len(t), "tile", hp.ctx.last_decode_tile())
names = ["prologue (headers, tables) -> barrier", "staging loads + LDS writes", "classify + barrier(s)", "row loop (issue)", "drain of wave 0's stores"]
d = np.diff(t[:, :6], axis=1).astype(np.float64)
life = (t[:, 5] - t[:, 0]).astype(np.float64)
# The counters of different CUs / XCCs are not synchronised, so only differences inside a workgroup mean
# something.  Tick rate: the CUs' workgroup slots are busy for the whole kernel, so the summed lives equal
# (resident workgroups) x (kernel time); resident = 256 CUs x floor(160 KiB / LDS per workgroup).
RES = int(os.environ.get("PHASE_RESIDENT_WGS", "768" if wl == "single" else "512"))
MHZ = float(life.sum()) / (RES * kern_us)
print("counter ticks per us:", round(MHZ, 2))
res = {"workload": wl, "workgroups": int(len(t)), "life_us_median": float(np.median(life) / MHZ), "phases": {}}
for i, n in enumerate(names):
    res["phases"][n] = {"median_us": round(float(np.median(d[:, i]) / MHZ), 2), "p90_us": round(float(np.percentile(d[:, i], 90) / MHZ), 2),
                        "share_of_life": round(float(d[:, i].sum() / life.sum()), 3)}
res["resident_workgroups_assumed"] = RES
xcc = t[:, 7]
res["workgroups_per_xcc"] = {int(k): int((xcc == k).sum()) for k in np.unique(xcc)}
print(json.dumps(res, indent=1))
