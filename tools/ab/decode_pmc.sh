#!/bin/bash
# SQ counters of the decode kernels (bench.py --workload $1), two PMC passes
R=$GRAFT_REPO_ROOT; WL=${1:-dual}; O=$R/gpurun_out/decode_pmc_$WL; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/p1 -o p -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu --no-loss-paths --no-extras --placement first > /dev/null 2> $O/err1.txt
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p2 -o p -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu --no-loss-paths --no-extras --placement first > /dev/null 2>> $O/err1.txt
python - <<PY
import csv, glob, collections
for d in ("p1", "p2"):
    f = glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        if "k_decode" not in k: continue
        acc[k.split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, c in acc.items():
        print(k, {a: round(sum(b) / len(b) / 1e6, 2) for a, b in c.items()}, "launches", len(next(iter(c.values()))))
PY
