#!/usr/bin/env python3
"""Timeline of k_decode_wide_fixup's tickets (instrumented library, tools/ab/phase_timing.sh build): stray10-like batch."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
buf = torch.zeros((1 << 12, 64), dtype=torch.int64, device="cuda")
import bench
N = 256
hp, base, out, *_ = bench._workload_setup("dual", N, pool_frames=8)
slots = base.shape[1]
packets = base
pk = packets.clone()
counts = torch.full((N,), slots, dtype=torch.int32, device="cuda")
f_idx = torch.arange(N, device="cuda")
lost = (f_idx * 7 + 3) % slots
keep = torch.arange(slots, device="cuda").unsqueeze(0).expand(N, slots)
keep = keep[keep != lost.unsqueeze(1)].reshape(N, slots - 1)
which = sys.argv[1] if len(sys.argv) > 1 else "stray10"
if which == "stray10":
    comp = f_idx[f_idx % 20 == 3]; swp = f_idx[f_idx % 20 == 13]
elif which == "compact1":
    comp = f_idx[:1]; swp = f_idx[:0]
else:
    comp = f_idx[:0]; swp = f_idx[:1]
if comp.numel():
    pk[comp, :slots - 1] = packets[comp.unsqueeze(1), keep[comp]]; pk[comp, slots - 1] = 0; counts[comp] = slots - 1
if swp.numel():
    a_, b_ = packets[swp, 10].clone(), packets[swp, 11].clone(); pk[swp, 10], pk[swp, 11] = b_, a_
for _ in range(10):
    hp.decode(pk, out, packet_counts=counts)
torch.cuda.synchronize()
os.environ["OUSTER_HIP_PHASE_BUF"] = hex(buf.data_ptr()); os.environ["OUSTER_HIP_PHASE_FIXUP_ONLY"] = "1"
buf.zero_(); torch.cuda.synchronize()
hp.decode(pk, out, packet_counts=counts)
torch.cuda.synchronize()
t = buf.cpu().numpy()
t = t[t[:, 0] != 0]
CLK = 2100.0  # cycles per us (approximately)
t0 = t[:, 0].min()
print(which, "workgroups", len(t), "kernel span %.1f us" % ((t[:, 63].max() - t0) / CLK))
print("first ticket start after kernel start: median %.1f us" % np.median([(r[3] - r[0]) / CLK for r in t if r[1] > 0]))
ev = []
for r in t:
    for e in range(int(r[1])):
        kind = int(r[2 + 4 * e]) & 0xff; n = int(r[2 + 4 * e]) >> 8
        ev.append((kind, n, (r[3 + 4 * e] - t0) / CLK, (r[4 + 4 * e] - r[3 + 4 * e]) / CLK if r[4 + 4 * e] else 0.0, (r[5 + 4 * e] - r[3 + 4 * e]) / CLK))
ev = np.array(ev)
res = ev[ev[:, 0] == 1]; redo = ev[ev[:, 0] == 2]
print("RESOLVE tickets %d: start median %.1f, duration median %.1f max %.1f us" % (len(res), np.median(res[:, 2]), np.median(res[:, 4]), res[:, 4].max()))
print("REDO tickets %d: start median %.1f us; wait for ready median %.1f max %.1f; tiles per ticket mean %.2f" %
      (len(redo), np.median(redo[:, 2]), np.median(redo[:, 3]), redo[:, 3].max(), redo[:, 1].mean()))
w = redo[redo[:, 1] > 0]
if len(w):
    print("  with work %d: (duration - wait) per tile median %.1f us p90 %.1f; end of last %.1f us" %
          (len(w), np.median((w[:, 4] - w[:, 3]) / w[:, 1]), np.percentile((w[:, 4] - w[:, 3]) / w[:, 1], 90), (w[:, 2] + w[:, 4]).max()))
nw = redo[redo[:, 1] == 0]
if len(nw):
    print("  without work %d: duration median %.1f us (wait %.1f)" % (len(nw), np.median(nw[:, 4]), np.median(nw[:, 3])))
print("tickets per workgroup: mean %.2f max %d" % (t[:, 1].mean(), t[:, 1].max()))
