#!/bin/bash
# rocprofv3 kernel trace of tools/ab/fixup_one.py: what do the two decode kernels take with K flagged frames?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fixup_prof; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
WL=${1:-dual}; KIND=${2:-swap}
for K in 0 1 16 26; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/k$K -o r -- python $R/tools/ab/fixup_one.py $WL $KIND $K > $O/k$K.log 2>&1
  f=$(find $O/k$K -name '*kernel_stats.csv' | head -1)
  echo "== $WL $KIND K=$K"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "decode" in r["Name"]:
        print("  %-60s calls %4s avg %8.1f us min %8.1f max %8.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
