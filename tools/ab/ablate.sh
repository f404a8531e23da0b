#!/bin/bash
# Experiment build: the product library with the XYZ arithmetic of the decode kernels replaced by a cast of the range
# (-DOUSTER_ABLATE_XYZ_MATH; every load and store stays) -> tools/ab/libouster_hip_noxyzmath.so, for ab_inproc.py:
#   python tools/ab/ab_inproc.py dual 256 base= nomath=tools/ab/libouster_hip_noxyzmath.so
# Run where hipcc is (the build container); the .so travels with the snapshot.
set -e
cd "$(dirname "$0")/../.."
B=/tmp/ouster_ablate; mkdir -p $B
F="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-result -DOUSTER_ABLATE_XYZ_MATH"
C=ouster_sdk_amd/csrc
for i in 0 1 2 3 4 5; do hipcc $F -DOUSTER_SPEC_ID=$i -c -o $B/k_decode_$i.o $C/k_decode.hip & done
for i in 1 2 3 4 5; do hipcc $F -DOUSTER_SPEC_ID=$i -c -o $B/k_decode_stream_$i.o $C/k_decode_stream.hip & done
hipcc $F -c -o $B/k_standalone.o $C/k_standalone.hip &
hipcc $F -c -o $B/ouster_hip_capi.o $C/ouster_hip_capi.hip &
wait
hipcc $F -shared -o tools/ab/libouster_hip_noxyzmath.so $B/*.o
ls -la tools/ab/libouster_hip_noxyzmath.so
