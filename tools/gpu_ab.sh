#!/bin/bash
# A/B of library builds on one box, interleaved.  usage: gpu_ab.sh <libA.so> <libB.so> [rounds]
R=$GRAFT_REPO_ROOT; cd $R
A=$1; B=$2; N=${3:-3}
P='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["kernel_ms_avg"])'
for i in $(seq $N); do
  OUSTER_HIP_OWN_STREAM=1 OUSTER_HIP_SO=$R/$A python bench.py --steps 40 --warmup 5 --no-cpu 2>/dev/null | python -c "$P" A:$A
  OUSTER_HIP_OWN_STREAM=1 OUSTER_HIP_SO=$R/$B python bench.py --steps 40 --warmup 5 --no-cpu 2>/dev/null | python -c "$P" B:$B
done
