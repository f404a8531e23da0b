import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from ouster_sdk_amd.device import HotPath
cal = O.synthetic_calib(h=128, w=2048, profile="RNG15_RFL8_NIR8_DUAL")
packets, src = O.synth_packets(cal, 4)
n = 64
host = np.concatenate([packets] * (n // 4))
hp = HotPath("RNG15_RFL8_NIR8_DUAL", 128, 2048, 16)
hp.set_pixel_shift_by_row(cal.pixel_shift_by_row)
hp.add_lut(cal.beam_to_lidar, cal.lut_transform(False), cal.beam_azimuth_angles, cal.beam_altitude_angles)
dev = torch.from_numpy(host).cuda()
out = hp.alloc_outputs(n, destagger=["RANGE", "RANGE2", "REFLECTIVITY", "REFLECTIVITY2"], xyz=["RANGE", "RANGE2"])
hp.decode(dev, out); hp.sync()
rng = out["RANGE"].cpu().numpy(); dst = out["destaggered:RANGE"].cpu().numpy()
want = np.stack([O.destagger(rng[k], cal.pixel_shift_by_row) for k in range(n)])
print("fused dst == oracle:", np.array_equal(dst, want))
if not np.array_equal(dst, want):
    bad = np.argwhere(dst != want); print(len(bad), bad[:8], np.unique(bad[:,0])[:16], np.unique(bad[:,1])[:16], np.unique(bad[:,2])[:32])
back = hp.destagger(out["destaggered:RANGE"], inverse=True)
bn = back.cpu().numpy()
print("back == RANGE:", np.array_equal(bn, rng), torch.equal(back, out["RANGE"]))
if not np.array_equal(bn, rng):
    bad = np.argwhere(bn != rng); print(len(bad), bad[:8], np.unique(bad[:,0])[:16], np.unique(bad[:,1])[:16], np.unique(bad[:,2])[:32])
    wb = np.stack([O.destagger(dst[k], cal.pixel_shift_by_row, True) for k in range(n)])
    print("oracle inverse of fused == RANGE:", np.array_equal(wb, rng), " standalone inverse == oracle inverse:", np.array_equal(bn, wb))
