#!/usr/bin/env python3
"""Find serialised memory round trips in compiled kernels: a load followed, a few instructions later, by a wait for ALL
outstanding loads (`s_waitcnt vmcnt(0)`), several times inside one loop body -- one full memory latency per piece.  (This is
how the twelve pose loads and the fifteen staging pieces of the fix-up tiles were found)
usage: hipcc --offload-arch=gfx950 -O3 -std=c++20 [-DOUSTER_SPEC_ID=n] --save-temps -c csrc/<file>.hip;  isa_waits.py <file>.s [min_chain]"""
import re
import sys
from collections import defaultdict

path = sys.argv[1]
min_chain = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kernel, loop = None, None
chains = defaultdict(list)   # (kernel, loop) -> [distance load->wait]
last_load = None
for n, line in enumerate(open(path), 1):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        kernel, loop, last_load = m.group(1), None, None
        continue
    if kernel is None:
        continue
    if ".amdhsa_kernel" in line:
        kernel = None
        continue
    m = re.search(r"Loop Header: Depth=(\d+)|in Loop: Header=(\w+) Depth=(\d+)", line)
    lab = re.match(r"^(\.LBB\w+):", line)
    if lab and "Loop Header" in line:
        loop = lab.group(1)
    elif m and m.group(2):
        loop = "." + m.group(2) if not m.group(2).startswith(".") else m.group(2)
    if re.search(r"\b(global_load|flat_load|buffer_load)", line):
        last_load = n
    if "s_waitcnt" in line and "vmcnt(0)" in line and last_load is not None and n - last_load <= 60 and loop:
        chains[(kernel, loop)].append(n - last_load)
        last_load = None
out = [(k, l, d) for (k, l), d in chains.items() if len(d) >= min_chain]
for k, l, d in sorted(out, key=lambda x: -len(x[2])):
    print(len(d), "waits in loop", l, "of", k[:110], "distances", d[:12])
