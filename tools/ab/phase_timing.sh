#!/bin/bash
# Where does a k_decode_wide workgroup spend its life?  Builds an instrumented copy of the library
# (-DOUSTER_PHASE_TIMING: s_memtime stamps at the phase boundaries of wave 0 of every workgroup) into
# tools/ab/_phase/ and runs tools/ab/phase_timing.py against it.  Experiment only; the product build has
# no trace of it.  Usage (build container): bash tools/ab/phase_timing.sh build ; (GPU box) ... run <workload>
set -e
cd "$(dirname "$0")/../.."
O=tools/ab/_phase
if [ "$1" = build ]; then
  mkdir -p $O
  F="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-result -DOUSTER_PHASE_TIMING"
  for i in 0 1 2 3 4 5; do hipcc $F -DOUSTER_SPEC_ID=$i -c -o $O/k_decode_$i.o ouster_sdk_amd/csrc/k_decode.hip & done
  for i in 1 2 3 4 5; do hipcc $F -DOUSTER_SPEC_ID=$i -c -o $O/k_decode_stream_$i.o ouster_sdk_amd/csrc/k_decode_stream.hip & done
  hipcc $F -c -o $O/k_standalone.o ouster_sdk_amd/csrc/k_standalone.hip &
  hipcc $F -c -o $O/capi.o ouster_sdk_amd/csrc/ouster_hip_capi.hip &
  wait
  hipcc $F -shared -o $O/libouster_hip_phase.so $O/*.o
  ls -la $O/libouster_hip_phase.so
else
  OUSTER_HIP_SO=$PWD/$O/libouster_hip_phase.so python tools/ab/phase_timing.py "${2:-single}"
fi
