#!/usr/bin/env python3
"""Phase timeline of k_decode_wide workgroups (instrumented library: tools/ab/build_variant.sh <name> -DOUSTER_PHASE_TIMING, run with
OUSTER_HIP_SO=tools/ab/libouster_hip_<name>.so).
  phase_timing.py <workload>        a large launch: phase shares per workgroup (per-CU cycle counters)
  phase_timing.py small [frames=1]  ONE small launch (1 or 4 frames) on the chip-wide 100 MHz clock (build with -DOUSTER_PHASE_WALL
                                    as well): when workgroups start, where each phase ends, when the fix-up kernel runs
  phase_timing.py fixup             the fix-up pass behind bench.py's stray10 batch (26 of 256 frames flagged) on the same clock
                                    (OUSTER_HIP_PHASE_FIXUP_ONLY=1 is set here): LEAD / REDO tickets, when they start, how long they take"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def small_launch(n):
    NWG = 1 << 14
    buf = torch.zeros((NWG, 16), dtype=torch.int64, device="cuda")
    os.environ["OUSTER_HIP_PHASE_BUF"] = hex(buf.data_ptr())
    import bench
    hp, packets, out, profile, shifts, lut_args, n_ret, _ = bench._workload_setup("fused4" if n == 4 else "dual", n)
    for _ in range(30):
        hp.decode(packets, out)
    torch.cuda.synchronize()
    rows = []
    for rep in range(20):
        buf.zero_()
        torch.cuda.synchronize()
        hp.decode(packets, out)
        torch.cuda.synchronize()
        t = buf.cpu().numpy()
        fix = t.reshape(-1, 64)[:, 62:64]
        fix = fix[fix[:, 0] != 0]
        wg = t[t[:, 0] != 0][:, :6].astype(np.float64)
        t0 = wg[:, 0].min()
        rel = (wg - t0) / 100.0   # us
        rows.append({"wgs": len(wg), "start_last": rel[:, 0].max(), "p1_med": np.median(rel[:, 1]), "p2_med": np.median(rel[:, 2]), "p3_med": np.median(rel[:, 3]),
                     "p4_med": np.median(rel[:, 4]), "p4_max": rel[:, 4].max(), "drain_max": rel[:, 5].max(),
                     "life_med": np.median(rel[:, 5] - rel[:, 0]),
                     "d01": np.median(rel[:, 1] - rel[:, 0]), "d12": np.median(rel[:, 2] - rel[:, 1]), "d23": np.median(rel[:, 3] - rel[:, 2]),
                     "d34": np.median(rel[:, 4] - rel[:, 3]), "d45": np.median(rel[:, 5] - rel[:, 4]),
                     "fix_start": (fix[:, 0].min() - t0) / 100.0 if len(fix) else None, "fix_end": (fix[:, 1].max() - t0) / 100.0 if len(fix) else None})
    keys = rows[0].keys()
    med = {k: round(float(np.median([r[k] for r in rows if r[k] is not None])), 2) for k in keys}
    print(json.dumps({"frames": n, "kernel": hp.ctx.last_decode_kernel(), "tile": hp.ctx.last_decode_tile(), "median_over_20_launches_us": med}))


def fixup_pass():
    os.environ["OUSTER_HIP_PHASE_FIXUP_ONLY"] = "1"
    NWG = 1 << 12
    buf = torch.zeros((NWG, 64), dtype=torch.int64, device="cuda")
    os.environ["OUSTER_HIP_PHASE_BUF"] = hex(buf.data_ptr())
    import bench
    F = 256
    hp, packets, out, profile, shifts, lut_args, n_ret, _ = bench._workload_setup("dual", F)
    for k, v in [kv.split("=") for kv in sys.argv[2:]]:
        hp.ctx.set_knob(k, int(v))
    slots = packets.shape[1]
    f_idx = torch.arange(F, device="cuda")
    lost = (f_idx * 7 + 3) % slots
    keep = torch.arange(slots, device="cuda").unsqueeze(0).expand(F, slots)
    keep = keep[keep != lost.unsqueeze(1)].reshape(F, slots - 1)
    pk = packets.clone()
    counts = torch.full((F,), slots, dtype=torch.int32, device="cuda")
    comp = f_idx[f_idx % 20 == 3]
    pk[comp, :slots - 1] = packets[comp.unsqueeze(1), keep[comp]]
    pk[comp, slots - 1] = 0
    counts[comp] = slots - 1
    swp = f_idx[f_idx % 20 == 13]
    a, b = packets[swp, 10].clone(), packets[swp, 11].clone()
    pk[swp, 10], pk[swp, 11] = b, a
    for _ in range(30):
        hp.decode(pk, out, packet_counts=counts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hp.decode(pk, out, packet_counts=counts)
    e1.record()
    torch.cuda.synchronize()
    step_ms = e0.elapsed_time(e1) / 20
    runs = []
    for rep in range(10):
        buf.zero_()
        torch.cuda.synchronize()
        hp.decode(pk, out, packet_counts=counts)
        torch.cuda.synchronize()
        t = buf.cpu().numpy().astype(np.float64)
        t = t[t[:, 62] != 0]
        k0 = t[:, 62].min()
        ev = {"lead": [], "redo_work": [], "redo_idle": []}
        for w in t:
            for e in range(int(w[1])):
                kind, n = int(w[2 + 4 * e]) & 0xff, int(w[2 + 4 * e]) >> 8
                b0, mid, end = (w[3 + 4 * e] - k0) / 100, (w[4 + 4 * e] - k0) / 100, (w[5 + 4 * e] - k0) / 100
                ev["lead" if kind == 1 else ("redo_work" if n else "redo_idle")].append((b0, end - b0, n))
        def summ(v):
            if not v: return None
            st, du = np.array([x[0] for x in v]), np.array([x[1] for x in v])
            return {"n": len(v), "start_med": round(float(np.median(st)), 1), "start_max": round(float(st.max()), 1), "dur_med": round(float(np.median(du)), 1),
                    "dur_max": round(float(du.max()), 1), "end_max": round(float((st + du).max()), 1), "tiles": int(sum(x[2] for x in v))}
        pct = lambda v: [round(float(x) / 100, 1) for x in np.percentile(v, [0, 25, 50, 75, 90, 100])]
        ws = sorted(x[0] for x in ev["redo_work"])
        runs.append({"wgs": len(t), "kernel_entry_pct_us": pct(t[:, 62] - k0), "crew_start_pct_us": pct(t[:, 0] - k0), "wg_exit_pct_us": pct(t[:, 63] - k0),
                     "kernel_end": round(float((t[:, 63] - k0).max()) / 100, 1), "redo_work_start_pct_us": [round(float(x), 1) for x in np.percentile(ws, [0, 25, 50, 75, 90, 100])],
                     **{k: summ(v) for k, v in ev.items()}})
    runs.sort(key=lambda r: r["kernel_end"])
    print(json.dumps({"step_ms": round(step_ms, 4), "tile": hp.ctx.last_decode_tile(), "kernel": hp.ctx.last_decode_kernel(), "median_run_us": runs[len(runs) // 2]}))


if len(sys.argv) > 1 and sys.argv[1] == "fixup":
    fixup_pass()
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "small":
    small_launch(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    sys.exit(0)
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
