#!/bin/bash
# memory-side counters of the decode kernels (bench.py --workload $1): is the L2 <-> fabric interface
# (EA) back-pressuring, or are the CUs not issuing enough?  Separate --pmc passes (4 TCC counters each).
R=$GRAFT_REPO_ROOT; WL=${1:-dual}; O=$R/gpurun_out/mem_pmc_$WL; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_BUSY_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_CYCLE_sum" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu > /dev/null 2> $O/err$i.txt || tail -3 $O/err$i.txt
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/p*")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        if "k_decode" not in k or "fixup" in k: continue
        acc[k.split("(")[0][-48:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, c in acc.items():
        print(k, {a: round(sum(b) / len(b) / 1e6, 3) for a, b in c.items()}, "launches", len(next(iter(c.values()))))
PY
