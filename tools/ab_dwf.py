import sys, os, json, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ouster_sdk_amd.device import HotPath
from tools.bench_kernels import timeit
H, W, N = 128, 2048, 256
alt, az, shifts, b2l, l2s = bench.synth_calibration()
hp = HotPath("RNG15_RFL8_NIR8_DUAL", H, W, 16)
lut = hp.add_lut(b2l, l2s, az, alt)
rng = torch.randint(0, 2 ** 19, (N, H, W), dtype=torch.int64, device="cuda")
rz = (rng * (torch.rand(rng.shape, device="cuda") >= 0.3)).to(torch.uint32)
status = torch.ones((N, W), dtype=torch.int32, device="cuda").to(torch.uint32)
poses = torch.eye(4, dtype=torch.float64, device="cuda").repeat(N, W, 1, 1).contiguous()
res = {}
for name, kw in (("full", {}), ("no_stores(capacity=1)", {"capacity": 1}), ("gate_nothing", {"gate": (600.0, 700.0)}),
                 ("keep_all", {"gate": (0.0, 1000.0)})):
    lo, hi = kw.pop("gate", (0.5, 400.0))
    s = timeit(lambda: hp.dewarp_frames(rz, status, poses, lo, hi, provenance=False, **kw))
    res[name] = round(s * 1e3, 3)
print(json.dumps(res))
