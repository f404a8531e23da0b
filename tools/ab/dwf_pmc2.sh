#!/bin/bash
# PMC passes over the three-kernel frame dewarp (tools/ab/dwf_only.py): where do k_dwf_emit's wave cycles go?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dwf_pmc2; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/tools/ab/dwf_only.py $1 > $O/run.json 2> $O/err.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU" \
           "TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_64B_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_$i -o p -- python $R/tools/ab/dwf_only.py $1 > /dev/null 2>> $O/err.txt
done
python - <<PY
import csv, glob, collections
print(open("$O/run.json").read().strip())
f = glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True)
for row in list(csv.DictReader(open(f[0])))[:8]:
    if "dwf" in row["Name"]: print("  ", row["Name"][:70], row["Calls"], row["AverageNs"])
for d in sorted(glob.glob("$O/pmc_*")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no output"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        if "dwf_emit" not in k: continue
        acc[k[:40]][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, c in acc.items():
        print("  ", k, {a: round(b / 10, 0) for a, b in c.items()})
PY
