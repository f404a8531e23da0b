#!/bin/bash
# does the variant tuner pick the variant that is fastest in steady state?  (same build, same box)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
o=gpurun_out/tuner_check.txt
: > $o
fmt='import sys,json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    r=j["roofline"]; print(j["ms_per_step"], r["kernel"], r["kernel_ms_avg"], r["frac"])'
for wl in ${@:-single dual}; do
  for v in tuner tuner 256 128 0; do
    echo "== $wl $v" >> $o
    if [ $v = tuner ]; then python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu 2>/dev/null | python -c "$fmt" >> $o
    else OUSTER_HIP_WIDE=$v python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu 2>/dev/null | python -c "$fmt" >> $o; fi
  done
done
cat $o
