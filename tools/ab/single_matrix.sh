#!/bin/bash
# A/B matrix for the 12 B/px profile (configs[1]): forced tile variants of the same build, same box
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
o=gpurun_out/single_matrix.txt
: > $o
run() { echo "== $*" >> $o; env "$@" python bench.py --workload single --steps 30 --warmup 3 --no-cpu 2>>gpurun_out/single_matrix.err | python -c "
import sys,json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    r=j['roofline']; print(j['value'], j['ms_per_step'], r['kernel'], r['kernel_ms_avg'], r['frac'], r['box_d2d_copy_GBps'])
" >> $o; }
run X=1
run OUSTER_HIP_WIDE=0
run OUSTER_HIP_WIDE=128 OUSTER_HIP_WIDE_KB=48
run OUSTER_HIP_WIDE=256 OUSTER_HIP_WIDE_KB=48
run OUSTER_HIP_WIDE=128 OUSTER_HIP_WIDE_KB=64
run OUSTER_HIP_WIDE=256 OUSTER_HIP_WIDE_KB=64
run OUSTER_HIP_WIDE=256 OUSTER_HIP_WIDE_KB=96
run OUSTER_HIP_WIDE=128 OUSTER_HIP_WIDE_KB=24
echo "== dual default" >> $o
python bench.py --steps 30 --warmup 3 --no-cpu 2>>gpurun_out/single_matrix.err >> $o
cat $o
