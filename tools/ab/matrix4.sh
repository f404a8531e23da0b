#!/bin/bash
cd "$(dirname "$0")/../.."
wl=${1:-single}
o=gpurun_out/matrix4_$wl.txt
: > $o
run() { echo "== $*" >> $o; env "$@" python bench.py --workload $wl --steps 40 --warmup 3 --no-cpu 2>>gpurun_out/matrix.err | python -c "
import sys,json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    r=j['roofline']; print(j['value'], j['ms_per_step'], r['kernel'], r['kernel_ms_avg'], r['frac'], r['frac_step'], r['box_d2d_copy_GBps'])
" >> $o; }
run OUSTER_HIP_WIDE=128
run OUSTER_HIP_WIDE=0
run OUSTER_HIP_WIDE=128 OUSTER_HIP_WIDE_ROWS=43 OUSTER_HIP_WIDE_KB=80
run OUSTER_HIP_WIDE=128 OUSTER_HIP_WIDE_ROWS=48 OUSTER_HIP_WIDE_KB=80
run OUSTER_HIP_WIDE=128 OUSTER_HIP_WIDE_ROWS=64 OUSTER_HIP_WIDE_KB=100
run OUSTER_HIP_WIDE=256 OUSTER_HIP_WIDE_ROWS=22 OUSTER_HIP_WIDE_KB=80
run OUSTER_HIP_WIDE=256 OUSTER_HIP_WIDE_ROWS=26 OUSTER_HIP_WIDE_KB=80
run OUSTER_HIP_WIDE=64 OUSTER_HIP_WIDE_ROWS=86 OUSTER_HIP_WIDE_KB=80
run OUSTER_HIP_WIDE=64 OUSTER_HIP_WIDE_ROWS=128 OUSTER_HIP_WIDE_KB=100
run OUSTER_HIP_WIDE=128
cat $o
