#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory into the summary committed under profiles/."""
import csv
import json
import os
import re
import sys

d = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "dual"


def rows(path):
    return list(csv.DictReader(open(path))) if os.path.exists(path) else []


print(f"# rocprofv3 summary ({os.path.basename(d)})\n")
print(f"Command: `python bench.py --workload {workload} --steps 20 --warmup 3 --no-cpu` (kernel trace); PMC passes use "
      "`--steps 3 --warmup 1`.\n")
print("## kernel stats (rocprofv3 --kernel-trace --stats)\n")
print("| kernel | calls | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|")
for r in rows(os.path.join(d, "trace", "bench_kernel_stats.csv"))[:8]:
    name = r["Name"].split("(")[0][-70:]
    print(f"| `{name}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
          f"{float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
# the timed region alone: bench.py's setup (variant tuner, placement draws) launches the decode kernels many
# times before the K timed steps, so the per-kernel average above covers all of it; the last K dispatches of
# the dominant kernel are the timed steps and must agree with the HIP-event average bench.py prints
tr = rows(os.path.join(d, "trace", "bench_kernel_trace.csv"))
try:
    steps = json.loads(open(os.path.join(d, "bench_under_rocprof.json")).read().strip().splitlines()[-1])["steps"]
except Exception:
    steps = 20
dec = [r for r in tr if "k_decode" in r["Kernel_Name"] and "fixup" not in r["Kernel_Name"]]
dec.sort(key=lambda r: int(r["Start_Timestamp"]))
if len(dec) >= steps:
    last = dec[-steps:]
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last]
    names = sorted({r["Kernel_Name"].split("(")[0][-70:] for r in last})
    print(f"\n**Timed region (the last {steps} decode dispatches of the trace):** `{'`, `'.join(names)}` "
          f"avg {sum(durs)/len(durs):.1f} us, min {min(durs):.1f}, max {max(durs):.1f} "
          f"(all {len(dec)} decode dispatches of the process: avg "
          f"{sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in dec)/len(dec):.1f} us -- "
          "setup draws several buffer placements and kernel variants, DESIGN.md 3.2b / 3.2c)")
print("\n## HBM traffic per launch (separate --pmc passes)\n")
agg = {}
for c in ("pmc_fetch", "pmc_write"):
    for r in rows(os.path.join(d, c, "bench_counter_collection.csv")):
        k = (r["Kernel_Name"].split("(")[0][-75:], r["Counter_Name"])
        agg.setdefault(k, []).append(float(r["Counter_Value"]))
print("| kernel | counter | launches | avg KB (raw) | MB corrected |")
print("|---|---|---|---|---|")
tot = {}
for (k, c), v in sorted(agg.items()):
    if "ouster_hip" not in k:
        continue
    avg = sum(v) / len(v)
    corr = avg * 1024 / 1e6 * (2.0 if c == "FETCH_SIZE" else 1.0)
    tot.setdefault(k, 0.0)
    tot[k] += corr
    print(f"| `{k}` | {c} | {len(v)} | {avg:.1f} | {corr:.1f} |")
print("\nFETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read stream, "
      "MI355X_MICROARCH.md; re-checked here on a torch 545 MB copy). Units: counters in KB.\n")
# machine-readable per-launch traffic of the dominant kernel, read back by bench.py (roofline.traffic)
# (the variant the tuner settled on = the decode kernel with the most launches)
dom = None
for (k, c), v in agg.items():
    if "k_decode" in k and "fixup" not in k and c == "WRITE_SIZE" and (dom is None or len(v) > len(agg[(dom, c)])):
        dom = k
parts = {}
for (k, c), v in agg.items():
    if k == dom:
        parts[c] = sum(v) / len(v) * 1024 * (2.0 if c == "FETCH_SIZE" else 1.0)
variants, variants_by_kernel = {}, {}
for (k, c), v in agg.items():
    m = re.search(r"(k_decode(?:_wide|_stream2?)?)<[^,]+, (\d+),", k)
    if m:
        kern = "k_decode_stream" if m.group(1).startswith("k_decode_stream") else m.group(1)
        key = "fetch_bytes" if c == "FETCH_SIZE" else "write_bytes"
        val = round(sum(v) / len(v) * 1024 * (2.0 if c == "FETCH_SIZE" else 1.0))
        variants_by_kernel.setdefault(f"{kern}:{m.group(2)}", {"kernel": k.strip(), "fetch_bytes": 0, "write_bytes": 0})[key] = val
        if kern != "k_decode_stream":
            variants.setdefault(m.group(2), {"kernel": k.strip(), "fetch_bytes": 0, "write_bytes": 0})[key] = val
for v in list(variants.values()) + list(variants_by_kernel.values()):
    v["total_bytes"] = v["fetch_bytes"] + v["write_bytes"]
if len(parts) == 2:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    json.dump({"kernel": dom.strip(), "workload": workload, "frames_per_launch": 256, "variants_by_tile_columns": variants,
               "variants": variants_by_kernel, "kernel_sources_sha256": bench.kernel_sources_sha256(),
               "fetch_bytes": round(parts["FETCH_SIZE"]), "write_bytes": round(parts["WRITE_SIZE"]),
               "total_bytes": round(parts["FETCH_SIZE"] + parts["WRITE_SIZE"]),
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of "
                         "bench.py --steps 3 --warmup 1; counters in KB; FETCH_SIZE x2 (gfx950)"},
              open(os.path.join(d, "pmc_traffic.json"), "w"), indent=1)
for name in ("bench_under_rocprof.json", "bench_plain.json"):
    p = os.path.join(d, name)
    if os.path.exists(p) and os.path.getsize(p):
        try:
            j = json.loads(open(p).read().strip().splitlines()[-1])
            rf = j["roofline"]
            print(f"## {name}\n\nvalue {j['value']} Mpoints/s, ms/step {j['ms_per_step']}, k_decode avg "
                  f"{rf['kernel_ms_avg']} ms (HIP events), achieved {rf['achieved']} GB/s = {rf['frac']*100:.1f} % "
                  f"of 8 TB/s; algorithmic {rf['algorithmic_bytes_per_launch']/1e6:.1f} MB/launch; "
                  f"box d2d copy {rf.get('box_d2d_copy_GBps')} GB/s\n")
            for k, t in tot.items():
                if "k_decode" in k:
                    print(f"`{k.strip()}` HBM traffic (PMC) {t:.1f} MB/launch = "
                          f"{t*1e6/rf['algorithmic_bytes_per_launch']:.3f} x algorithmic\n")
            if j.get("cpu_baseline"):
                print(f"cpu_baseline: {json.dumps(j['cpu_baseline'])}\n")
        except Exception as e:
            print(f"({name}: {e})")
