#!/bin/bash
# non-temporal hints on the tile loads / on the output stores of the decode kernels: experiment builds of
# the library in tools/ab/_nt/ (build: in the container), same-box A/B against the product build (run).
set -e
cd "$(dirname "$0")/../.."
O=tools/ab/_nt
if [ "$1" = build ]; then
  mkdir -p $O
  for v in ${NT_VARIANTS:-"ld:-DOUSTER_NT_LOADS=1" "st:-DOUSTER_NT_STORES=1" "ldst:-DOUSTER_NT_LOADS=1 -DOUSTER_NT_STORES=1"}; do
    tag=${v%%:*}; defs=${v#*:}
    F="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-result $defs"
    mkdir -p $O/$tag
    for i in 0 1 2 3 4 5; do hipcc $F -DOUSTER_SPEC_ID=$i -c -o $O/$tag/k_decode_$i.o ouster_sdk_amd/csrc/k_decode.hip & done
    hipcc $F -c -o $O/$tag/k_standalone.o ouster_sdk_amd/csrc/k_standalone.hip &
    hipcc $F -c -o $O/$tag/capi.o ouster_sdk_amd/csrc/ouster_hip_capi.hip &
    wait
    hipcc $F -shared -o $O/libouster_hip_nt_$tag.so $O/$tag/*.o
    rm -rf $O/$tag
  done
  ls -la $O
else
  fmt='import sys,json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    r=j["roofline"]; print(r["kernel"], r["kernel_ms_avg"], r["frac"])'
  for wl in dual single; do
    for rep in 1 2; do
      for tag in base ld st ldst; do
        echo "== $wl $tag #$rep"
        if [ $tag = base ]; then OUSTER_HIP_WIDE=${WIDE:-128} python bench.py --workload $wl --steps 60 --warmup 3 --no-cpu 2>/dev/null | python -c "$fmt"
        else OUSTER_HIP_WIDE=${WIDE:-128} OUSTER_HIP_SO=$PWD/$O/libouster_hip_nt_$tag.so python bench.py --workload $wl --steps 60 --warmup 3 --no-cpu 2>/dev/null | python -c "$fmt"; fi
      done
    done
  done
fi
