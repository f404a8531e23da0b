#!/bin/bash
# Experiment build of the HIP library with extra compiler flags -> tools/ab/libouster_hip_<name>.so (for ab_inproc.py).
#   tools/ab/build_variant.sh maxilp -mllvm -amdgpu-sched-strategy=max-ilp
# Run where hipcc is (the build container); the .so travels with the snapshot.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../.."
B=/tmp/ouster_variant_$NAME; mkdir -p $B
F="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-result $*"
C=ouster_sdk_amd/csrc
for i in 0 1 2 3 4 5; do hipcc $F -DOUSTER_SPEC_ID=$i -c -o $B/k_decode_$i.o $C/k_decode.hip & done
for i in 1 2 3 4 5; do hipcc $F -DOUSTER_SPEC_ID=$i -c -o $B/k_decode_stream_$i.o $C/k_decode_stream.hip & done
hipcc $F -c -o $B/k_standalone.o $C/k_standalone.hip &
hipcc $F -c -o $B/ouster_hip_capi.o $C/ouster_hip_capi.hip &
hipcc $F -c -o $B/host_pool.o $C/host_pool.hip &
wait
hipcc $F -shared -o tools/ab/libouster_hip_$NAME.so $B/*.o
ls -la tools/ab/libouster_hip_$NAME.so
