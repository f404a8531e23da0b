#!/bin/bash
# rocprofv3 kernel trace of bench.py for the current build and the r01 library (same box)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_ab; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for wl in dual single; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/new_$wl -o bench -- python $R/bench.py --workload $wl --steps 30 --warmup 3 --no-cpu > $O/new_$wl.json 2> $O/new_$wl.err
  OUSTER_HIP_SO=$R/tools/ab/libouster_hip_r01.so rocprofv3 --kernel-trace --stats --output-format csv -d $O/r01_$wl -o bench -- python $R/bench.py --workload $wl --steps 30 --warmup 3 --no-cpu > $O/r01_$wl.json 2> $O/r01_$wl.err
done
cd $O
for d in new_dual r01_dual new_single r01_single; do echo "== $d"; f=$(find $d -name '*kernel_stats.csv' | head -1); head -8 $f | cut -c1-200; done
