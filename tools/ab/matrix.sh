#!/bin/bash
# tile-variant matrix of the current build on one box: bench.py --workload $1 with forced variants
cd "$(dirname "$0")/../.."
wl=${1:-single}
mkdir -p gpurun_out
o=gpurun_out/matrix_$wl.txt
: > $o
run() { echo "== $*" >> $o; env "$@" python bench.py --workload $wl --steps 40 --warmup 3 --no-cpu 2>>gpurun_out/matrix.err | python -c "
import sys,json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    r=j['roofline']; print(j['value'], j['ms_per_step'], r['kernel'], r['kernel_ms_avg'], r['frac'], r['frac_step'], r['box_d2d_copy_GBps'])
" >> $o; }
run X=1
run OUSTER_HIP_WIDE=0
run OUSTER_HIP_WIDE=0 OUSTER_HIP_TILE=16
run OUSTER_HIP_WIDE=0 OUSTER_HIP_TILE=32
run OUSTER_HIP_WIDE=0 OUSTER_HIP_TILE=64
run OUSTER_HIP_WIDE=64 OUSTER_HIP_WIDE_KB=52
run OUSTER_HIP_WIDE=64 OUSTER_HIP_WIDE_KB=32
run OUSTER_HIP_WIDE=128 OUSTER_HIP_WIDE_KB=52
run OUSTER_HIP_WIDE=128 OUSTER_HIP_WIDE_KB=32
run OUSTER_HIP_WIDE=256 OUSTER_HIP_WIDE_KB=52
run OUSTER_HIP_WIDE=256 OUSTER_HIP_WIDE_KB=32
run OUSTER_HIP_WIDE=128 OUSTER_HIP_WIDE_KB=72
run OUSTER_HIP_WIDE=256 OUSTER_HIP_WIDE_KB=72
run OUSTER_HIP_WIDE=512 OUSTER_HIP_WIDE_KB=52
run X=2
cat $o
