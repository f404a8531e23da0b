import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from ouster_sdk_amd.device import HotPath
cal = O.synthetic_calib(h=128, w=2048, profile="RNG15_RFL8_NIR8_DUAL")
hp = HotPath("RNG15_RFL8_NIR8_DUAL", 128, 2048, 16)
hp.set_pixel_shift_by_row(cal.pixel_shift_by_row)
rng = np.random.default_rng(0)
x = rng.integers(0, 2**32, size=(64, 128, 2048), dtype=np.uint64).astype(np.uint32)
dx = torch.from_numpy(x).cuda()
d = hp.destagger(dx)
b = hp.destagger(d, inverse=True)
dn, bn = d.cpu().numpy(), b.cpu().numpy()
want = np.stack([O.destagger(x[k], cal.pixel_shift_by_row) for k in range(64)])
print("fwd equal oracle:", np.array_equal(dn, want), "roundtrip:", np.array_equal(bn, x), "torch.equal:", torch.equal(b, dx))
if not np.array_equal(bn, x):
    bad = np.argwhere(bn != x); print(len(bad), bad[:10])
if not np.array_equal(dn, want):
    bad = np.argwhere(dn != want); print(len(bad), bad[:10], np.unique(bad[:,0])[:10], np.unique(bad[:,1])[:20])
