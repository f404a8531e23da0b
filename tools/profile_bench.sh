#!/bin/bash
# rocprofv3 evidence for bench.py: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in
# separate PMC passes (as MI355X_MICROARCH.md prescribes).  Output: gpurun_out/prof_<tag>/
TAG=${1:-r02}
WL=${2:-dual}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --workload $WL --steps 20 --warmup 3 --no-cpu --no-loss-paths --no-extras > $O/bench_under_rocprof.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu --placement first --no-loss-paths --no-extras > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu --placement first --no-loss-paths --no-extras > /dev/null 2> $O/pmc_write.err
# the counters of THIS run become profiles/<tag>/pmc_traffic.json before the plain bench line is taken, so that its
# roofline.traffic cites them (recorded on this tree's kernel sources) and not an older directory's
cd $R; python tools/summarize_profile.py $O $WL > /dev/null 2>&1 || true
mkdir -p profiles/$TAG; cp $O/pmc_traffic.json profiles/$TAG/pmc_traffic.json
python bench.py --workload $WL --steps 20 --warmup 3 > $O/bench_plain.json 2> $O/bench_plain.err
python tools/summarize_profile.py $O $WL > $O/SUMMARY.md; cat $O/SUMMARY.md
