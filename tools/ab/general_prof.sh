#!/bin/bash
# rocprofv3 kernel trace of the general mapping (every frame compacted into W/cpp - 1 slots): k_slotmap + k_decode_wide from the maps
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/general_prof; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
cat > /tmp/general_one.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
wl = sys.argv[1]
N = 256
hp, packets, out, *_ = bench._workload_setup(wl, N, pool_frames=8)
slots = packets.shape[1]
f_idx = torch.arange(N, device="cuda")
lost = (f_idx * 7 + 3) % slots
keep = torch.arange(slots, device="cuda").unsqueeze(0).expand(N, slots)
keep = keep[keep != lost.unsqueeze(1)].reshape(N, slots - 1)
pk = packets[f_idx.unsqueeze(1), keep].contiguous()
counts = torch.full((N,), slots - 1, dtype=torch.int32, device="cuda")
for _ in range(30):
    hp.decode(pk, out, packet_counts=counts)
torch.cuda.synchronize()
PY
for wl in dual single; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$wl -o r -- python /tmp/general_one.py $wl > $O/$wl.log 2>&1
  f=$(find $O/$wl -name '*kernel_stats.csv' | head -1)
  echo "== $wl"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "decode" in r["Name"] or "slotmap" in r["Name"]:
        print("  %-70s calls %4s avg %8.1f us min %8.1f max %8.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
