#!/bin/bash
# same-box A/B: the round-1 library (tools/ab/libouster_hip_r01.so) against the current build
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
o=gpurun_out/ab_bench.txt
: > $o
fmt='import sys,json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    r=j["roofline"]; print(j["value"], "ms_step", j["ms_per_step"], r["kernel"], "kern_ms", r["kernel_ms_avg"], "frac", r["frac"], "frac_step", r.get("frac_step"), "d2d", r["box_d2d_copy_GBps"], "valid", j.get("validated"))'
for wl in dual single; do
  for rep in 1 2; do
    echo "== $wl ${AB_LIB:-r01} #$rep" >> $o
    OUSTER_HIP_SO=$PWD/tools/ab/${AB_LIB:-libouster_hip_r01.so} python bench.py --workload $wl --steps 100 --warmup 5 --no-cpu 2>>gpurun_out/ab.err | python -c "$fmt" >> $o
    echo "== $wl new #$rep" >> $o
    python bench.py --workload $wl --steps 100 --warmup 5 --no-cpu 2>>gpurun_out/ab.err | python -c "$fmt" >> $o
  done
done
cat $o
