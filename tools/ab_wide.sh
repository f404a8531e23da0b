#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for cfg in "tuned" "0" "128" "256" "tuned"; do
  if [ $cfg = tuned ]; then unset OUSTER_HIP_WIDE; else export OUSTER_HIP_WIDE=$cfg; fi
  python bench.py --steps 30 --warmup 5 --no-cpu 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], 'kernel', d['roofline']['kernel_ms_avg'], 'step', d['ms_per_step'], d['roofline']['kernel'])"
done
unset OUSTER_HIP_WIDE
python bench.py --steps 20 --warmup 3 --no-cpu --workload single 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single tuned', d['value'], 'kernel', d['roofline']['kernel_ms_avg'], d['roofline']['achieved'], d['roofline']['kernel'])"
OUSTER_HIP_WIDE=0 python bench.py --steps 20 --warmup 3 --no-cpu --workload single 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single narrow', d['value'], 'kernel', d['roofline']['kernel_ms_avg'], d['roofline']['achieved'], d['roofline']['kernel'])"
