#!/bin/bash
# tile-shape matrix after the prologue rework: smaller tiles = more workgroups per CU overlapping their staging
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
o=gpurun_out/matrix5.txt
: > $o
fmt='import sys,json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    r=j["roofline"]; print(r["kernel"], r["kernel_ms_avg"], r["frac"])'
run() { wl=$1; shift; echo "== $wl $*" >> $o; env "$@" python bench.py --workload $wl --steps 40 --warmup 3 --no-cpu 2>/dev/null | python -c "$fmt" >> $o; }
for rep in 1 2; do
run dual OUSTER_HIP_WIDE=128 OUSTER_HIP_WIDE_ROWS=64
run dual OUSTER_HIP_WIDE=128 OUSTER_HIP_WIDE_ROWS=32
run dual OUSTER_HIP_WIDE=128 OUSTER_HIP_WIDE_ROWS=40
run dual OUSTER_HIP_WIDE=256 OUSTER_HIP_WIDE_ROWS=32
run dual OUSTER_HIP_WIDE=256 OUSTER_HIP_WIDE_ROWS=16
run dual OUSTER_HIP_WIDE=256 OUSTER_HIP_WIDE_ROWS=20
run single OUSTER_HIP_WIDE=128 OUSTER_HIP_WIDE_ROWS=32
run single OUSTER_HIP_WIDE=128 OUSTER_HIP_WIDE_ROWS=24
run single OUSTER_HIP_WIDE=128 OUSTER_HIP_WIDE_ROWS=16
run single OUSTER_HIP_WIDE=256 OUSTER_HIP_WIDE_ROWS=16
run single OUSTER_HIP_WIDE=256 OUSTER_HIP_WIDE_ROWS=12
run single OUSTER_HIP_WIDE=64 OUSTER_HIP_WIDE_ROWS=64
done
cat $o
