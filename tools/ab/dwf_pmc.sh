#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dwf_pmc; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for v in 1 0; do
  OUSTER_HIP_DWF_SINGLE=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$v -o t -- python $R/tools/ab/dwf_only.py > $O/run_$v.json 2> $O/err_$v.txt
  OUSTER_HIP_DWF_SINGLE=$v rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/pmc_$v -o p -- python $R/tools/ab/dwf_only.py > /dev/null 2>> $O/err_$v.txt
  OUSTER_HIP_DWF_SINGLE=$v rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc2_$v -o p -- python $R/tools/ab/dwf_only.py > /dev/null 2>> $O/err_$v.txt
done
python - <<PY
import csv, glob, collections
for v in (1, 0):
    print("== single" if v else "== three-kernel", open("$O/run_%d.json" % v).read().strip())
    f = glob.glob("$O/trace_%d/**/*kernel_stats.csv" % v, recursive=True)
    for row in list(csv.DictReader(open(f[0])))[:6]:
        if "dwf" in row["Name"]: print("  ", row["Name"][:60], row["Calls"], row["AverageNs"])
    for d in ("pmc", "pmc2"):
        f = glob.glob("$O/%s_%d/**/*counter_collection.csv" % (d, v), recursive=True)
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for row in csv.DictReader(open(f[0])):
            k = row["Kernel_Name"]
            if "dwf" not in k: continue
            acc[k[:50]][row["Counter_Name"]] += float(row["Counter_Value"])
        for k, c in acc.items():
            print("  ", k, {a: round(b / 10, 0) for a, b in c.items()})
PY
