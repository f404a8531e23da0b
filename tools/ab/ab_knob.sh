#!/bin/bash
# same-box A/B of one environment knob: tools/ab/ab_knob.sh OUSTER_HIP_BEAM_LDS 0 1 [workloads...]
cd "$(dirname "$0")/../.."
K=$1; A=$2; B=$3; shift 3
o=gpurun_out/ab_knob.txt
: > $o
fmt='import sys,json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    r=j["roofline"]; print(j["value"], "ms_step", j["ms_per_step"], r["kernel"], "kern_ms", r["kernel_ms_avg"], "frac", r["frac"], "frac_step", r.get("frac_step"), "d2d", r["box_d2d_copy_GBps"])'
for wl in ${@:-dual single}; do
  for rep in 1 2; do
    for v in $A $B; do
      echo "== $wl $K=$v #$rep" >> $o
      env $K=$v python bench.py --workload $wl --steps 100 --warmup 5 --no-cpu 2>>gpurun_out/ab.err | python -c "$fmt" >> $o
    done
  done
done
cat $o
