#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "0 0" "128 0" "256 0" "128 1" "128 2" "0 0" "128 0"; do
  set -- $cfg
  OUSTER_HIP_WIDE=$1 OUSTER_HIP_DBG=$2 python bench.py --steps 30 --warmup 5 --no-cpu 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WIDE=$1 DBG=$2', 'kernel', d['roofline']['kernel_ms_avg'], 'step', d['ms_per_step'])"
done
