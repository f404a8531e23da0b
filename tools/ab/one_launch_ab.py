import sys, time, json
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import bench
res = {}
for n in (1, 4):
    for mode in ("two_launch", "no_fixup", "fused"):
        hp, packets, out, profile, shifts, lut_args, n_ret, _ = bench._workload_setup("fused4" if n == 4 else "dual", n)
        hp.ctx.set_knob("fused_tail", 1 if mode == "fused" else 0)
        hp.ctx.set_knob("fixup", 0 if mode == "no_fixup" else 1)
        inputs = [packets, packets.clone()]
        for _ in range(30):
            hp.decode(packets, out)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for i in range(400):
                hp.decode(inputs[i & 1], out)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 400)
        lat = []
        for i in range(100):
            t0 = time.perf_counter(); hp.decode(inputs[i & 1], out); torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
        res[f"{n}:{mode}"] = {"pipelined_us": round(best * 1e6, 2), "sync_us": round(float(np.median(lat)) * 1e6, 2), "kernel": hp.ctx.last_decode_kernel()}
print(json.dumps(res, indent=1))
