#!/bin/bash
# HBM traffic of the frame-dewarp kernels (tools/ab/dwf_only.py): FETCH_SIZE / WRITE_SIZE in separate passes, KB per launch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dwf_traffic; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/$c -o p -- python $R/tools/ab/dwf_only.py $1 > /dev/null 2>> $O/err.txt
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/tools/ab/dwf_only.py $1 > $O/run.json 2>> $O/err.txt
python - <<PY
import csv, glob, collections
print(open("$O/run.json").read().strip())
dur = {}
for row in csv.DictReader(open(glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True)[0])):
    if "dwf" in row["Name"]: dur[row["Name"].split("(")[0].split("<")[0].split("::")[-1]] = float(row["AverageNs"]) / 1e3
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for row in csv.DictReader(open(glob.glob("$O/%s/**/*counter_collection.csv" % c, recursive=True)[0])):
        k = row["Kernel_Name"]
        if "dwf" in k: acc[k.split("(")[0].split("<")[0].split("::")[-1]][c].append(float(row["Counter_Value"]))
for k, v in acc.items():
    f = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]) * 2 * 1024 / 1e6   # gfx950: FETCH_SIZE counts 2 KB units... (x2 per the guide)
    w = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"]) * 1024 / 1e6
    print("%-18s read %7.1f MB  write %7.1f MB  %6.1f us  -> %.2f TB/s" % (k, f, w, dur.get(k, 0), (f + w) / max(dur.get(k, 1), 1e-9) / 1e6 * 1e6 / 1e6))
PY
