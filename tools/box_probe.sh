#!/bin/bash
# correlate the box's clocks / power state / memory microbenchmarks with the bench result
cd $GRAFT_REPO_ROOT
O=gpurun_out/box_probe_$(date +%s); mkdir -p $O
rocm-smi --showclocks --showpower --showperflevel --showmemuse --showtemp --showmemvendor > $O/smi_before.txt 2>&1
rocminfo | grep -E "Marketing|Compute Unit|Max Clock|Name:.*gfx" | head -12 > $O/rocminfo.txt 2>&1
[ -x tools/membench ] || hipcc -O3 --offload-arch=gfx950 -o tools/membench tools/membench.hip 2> $O/membench_build.err
tools/membench > $O/membench.txt 2>&1
python bench.py --steps 40 --warmup 5 --no-cpu > $O/bench.json 2> $O/bench.err &
BP=$!
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20; do
  kill -0 $BP 2>/dev/null || break
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> $O/smi_during.txt; echo >> $O/smi_during.txt
  sleep 0.5
done
wait $BP
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("bench value", d["value"], "kernel_ms", d["roofline"]["kernel_ms_avg"], "achieved", d["roofline"]["achieved"], "d2d", d["roofline"]["box_d2d_copy_GBps"])
PY
for o in planes planes+dst xyz; do
  python bench.py --steps 30 --warmup 5 --no-cpu --outputs $o 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('outputs=$o kernel_ms', d['roofline']['kernel_ms_avg'], 'achieved', d['roofline']['achieved'])"
done
OUSTER_HIP_XCD=0 python bench.py --steps 30 --warmup 5 --no-cpu 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('XCD map off: kernel_ms', d['roofline']['kernel_ms_avg'], 'achieved', d['roofline']['achieved'])"
grep -E "sclk|mclk|fclk|Power|Perf" $O/smi_before.txt | head -4
echo "--- during (max sclk seen)"; grep -oE "sclk clock level: [0-9S]+: \([0-9]+Mhz\)" $O/smi_during.txt | sort -t'(' -k2 -n | tail -2; grep -oE "Power \(W\): [0-9.]+" $O/smi_during.txt | sort -t: -k2 -n | tail -1
tail -8 $O/membench.txt
