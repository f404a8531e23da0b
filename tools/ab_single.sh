#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in 0 64 32 0 64 32; do
  echo "ROWS=$t"; OUSTER_HIP_ROWS=$t python bench.py --workload single --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['achieved'])"
done
