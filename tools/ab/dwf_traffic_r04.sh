#!/bin/bash
# HBM traffic of the frame dewarp by route (tools/ab/dwf_route.py): FETCH_SIZE / WRITE_SIZE in separate passes + a kernel trace;
# writes gpurun_out/dwf_r04/dewarp_pmc.json (copied to profiles/r04_dewarp/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dwf_r04; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for route in counted own; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/$route/$c -o p -- python $R/tools/ab/dwf_route.py $route > /dev/null 2>> $O/err.txt
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$route/trace -o t -- python $R/tools/ab/dwf_route.py $route > $O/$route/run.json 2>> $O/err.txt
done
python3 - "$O" <<'PY'
import csv, glob, collections, json, sys
O = sys.argv[1]
res = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes + a --kernel-trace --stats pass of tools/ab/dwf_route.py; "
                 "values per launch; FETCH_SIZE in KB x2 (gfx950), WRITE_SIZE in KB", "routes": {}}
for route in ("counted", "own"):
    run = json.loads(open(f"{O}/{route}/run.json").read().strip().splitlines()[-1])
    dur = {}
    for row in csv.DictReader(open(glob.glob(f"{O}/{route}/trace/**/*kernel_stats.csv", recursive=True)[0])):
        if "dwf" in row["Name"]: dur[row["Name"].split("(")[0].split("<")[0].split("::")[-1]] = float(row["AverageNs"]) / 1e3
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for row in csv.DictReader(open(glob.glob(f"{O}/{route}/{c}/**/*counter_collection.csv", recursive=True)[0])):
            k = row["Kernel_Name"]
            if "dwf" in k: acc[k.split("(")[0].split("<")[0].split("::")[-1]][c].append(float(row["Counter_Value"]))
    kern, total = {}, 0
    for k, v in acc.items():
        f = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]) * 2 * 1024
        w = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"]) * 1024
        kern[k] = {"fetch_bytes": int(f), "write_bytes": int(w), "avg_us": round(dur.get(k, 0), 1),
                   "TBps": round((f + w) / max(dur.get(k, 1e-9), 1e-9) / 1e6, 2)}
        total += f + w
    res["routes"][route] = {"run": run, "kernels": kern, "total_bytes": int(total), "sum_of_kernels_us": round(sum(dur.values()), 1),
                            "traffic_over_algorithmic": round(total / run["algorithmic_bytes"], 3)}
json.dump(res, open(f"{O}/dewarp_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
