#!/bin/bash
# which hardware counters tell a fast allocation from a slow one?  (tools/ab/lottery_pmc.py under rocprofv3 --pmc)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lottery_pmc; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
i=0
# round 5: LOTTERY_CHANNELS=1 records the per-channel write / read request counters (one CSV row per TCC
# instance) instead: is a slow placement a channel imbalance?  LOTTERY_WORKLOAD=single LOTTERY_STREAM=128 for configs[1].
if [ -n "$LOTTERY_CHANNELS" ]; then
  i=0
  for set in "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" "TCC_EA0_RDREQ TCC_EA0_WRREQ_STALL" "TCC_EA0_WRREQ_DRAM TCC_EA0_RDREQ_DRAM"; do
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $set --kernel-include-regex "k_decode" --output-format json -d $O/c$i -o c -- python $R/tools/ab/lottery_pmc.py > $O/cout$i.txt 2> $O/cerr$i.txt || tail -3 $O/cerr$i.txt
    tail -1 $O/cout$i.txt
  done
  python $R/tools/ab/lottery_pmc.py --channels $O
  exit 0
fi
# round 6: LOTTERY_TRANSLATION=1 looks at the second level of address translation instead (UTCL2 busy cycles, first-level misses
# under a miss, stalls for UTCL2 credits): profiles/r06_latency/placement_notes.txt items 7 - 9
if [ -n "$LOTTERY_TRANSLATION" ]; then
  SETS=("GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
        "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_REQUEST_sum" \
        "TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_LFIFO_FULL_sum TCP_PENDING_STALL_CYCLES_sum")
else
  SETS=("TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" \
        "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum" \
        "GRBM_GUI_ACTIVE TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum")
fi
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- python $R/tools/ab/lottery_pmc.py > $O/out$i.txt 2> $O/err$i.txt || tail -3 $O/err$i.txt
  tail -1 $O/out$i.txt
done
python - <<PY
import csv, glob, json, collections
for d in sorted(glob.glob("$O/p*")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    rows = [r for r in csv.DictReader(open(f[0])) if "k_decode_wide" in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows: by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    info = json.loads(open(d.replace("/p", "/out") + ".txt").read().strip().splitlines()[-1])
    K, L = info["sets"], info["launches_per_set"]
    print("ms", info["ms_per_launch_under_the_profiler"])
    for c, v in by.items():
        v.sort(); vals = [x for _, x in v][-K * L:]
        print(c, [round(sum(vals[i * L:(i + 1) * L]) / L / 1e6, 3) for i in range(K)])
PY
