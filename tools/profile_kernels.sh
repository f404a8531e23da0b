#!/bin/bash
# rocprofv3 kernel trace of tools/bench_kernels.py (standalone kernels).  Output: gpurun_out/prof_kernels_<tag>/
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_kernels_$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o k -- python $R/tools/bench_kernels.py > $O/bench_kernels.json 2> $O/trace.err
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv; head -30 $O/kernel_stats.csv | cut -c1-200
