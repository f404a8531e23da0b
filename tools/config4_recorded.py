#!/usr/bin/env python3
"""BASELINE.json configs[3] with RECORDED frames: a batch of N (default 512) recorded OS-1-128 frames
that lives on rank 0 is scattered to the ranks (RCCL point-to-point over xGMI, one grouped
send per peer), decoded by every rank on its own GPU (fused decode + destagger + cartesian, no
collective), and the XYZ clouds are gathered back on rank 0.

  python tools/config4_recorded.py [--frames 512]                                  # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         tools/config4_recorded.py --frames 512                                    # 8 GPUs

The recorded frames come from the reference's own fixture capture (tests/golden/pcaps/
OS-2-128-U1_v2.3.0_1024x10.pcap: 128 beams, RNG19_RFL8_SIG16_NIR16 single return -- configs[3]'s
profile -- at 1024 columns, read with the C++ PcapReader).  The fixture holds 64 lidar packets that
straddle two frame ids, so the recorded frames are partial: every packet sits in its home slot, the
rest of the frame are holes, exactly what a lossy live stream looks like; they are cycled to fill
the batch.  Rank 0 prints one JSON line with the three stages
timed separately (the exchange is ~10x the sharded kernel time, so it is never folded
into the kernel rate) and a checksum of the gathered clouds that does not depend on the number of
ranks.  BENCH_ONE_DEVICE=1 / BENCH_BACKEND=gloo: every rank on cuda:0 / gloo instead of RCCL, to
exercise the N > 1 path on a 1-GPU box.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CAPTURE = os.path.join(ROOT, "tests", "golden", "pcaps", "OS-2-128-U1_v2.3.0_1024x10")


def read_sensor_json(path):
    """The handful of metadata fields the hot path needs (both JSON layouts of the fixtures)."""
    j = json.load(open(path))
    if "lidar_data_format" in j:
        df, bi = j["lidar_data_format"], j["beam_intrinsics"]
        l2s = j["lidar_intrinsics"]["lidar_to_sensor_transform"]
    else:
        df, bi, l2s = j["data_format"], j, j["lidar_to_sensor_transform"]
    b2l = np.eye(4)
    if "beam_to_lidar_transform" in bi:
        b2l = np.array(bi["beam_to_lidar_transform"], dtype=np.float64).reshape(4, 4)
    else:
        b2l[0, 3] = bi.get("lidar_origin_to_beam_origin_mm", 15.806)
    return {"h": df["pixels_per_column"], "w": df["columns_per_frame"], "cpp": df["columns_per_packet"],
            "profile": df["udp_profile_lidar"], "shifts": np.array(df["pixel_shift_by_row"], np.int32),
            "alt": np.array(bi["beam_altitude_angles"]), "az": np.array(bi["beam_azimuth_angles"]),
            "b2l": b2l, "l2s": np.array(l2s, dtype=np.float64).reshape(4, 4)}


def recorded_frames(pcap, packet_size, slots, cpp, header_frame_id=(2, 2)):
    """The capture's frames as [n, slots, packet_size]: every packet in its home slot, lost / not yet
    recorded packets left as holes (zeros = invalid columns)."""
    from ouster_sdk_amd import _capi
    L = _capi.load_core()
    L.ouster_pcap_read_udp.argtypes = [C.c_char_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                       C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
    cap = 64 << 20
    buf = np.zeros(cap, np.uint8)
    msg = C.create_string_buffer(256)
    trunc = C.c_int(0)
    n = L.ouster_pcap_read_udp(pcap.encode(), 0, packet_size, buf.ctypes.data, cap, None, None, 1 << 20, msg, 256,
                               C.byref(trunc))
    if n < 0:
        raise RuntimeError(msg.value.decode())
    pk = buf[: n * packet_size].reshape(n, packet_size)
    off, sz = header_frame_id
    fid = pk[:, off:off + sz].copy().view(np.uint16)[:, 0]
    mid = pk[:, 32 + 8:32 + 10].copy().view(np.uint16)[:, 0]   # first column's measurement id (STANDARD header)
    frames = []
    for f in np.unique(fid):
        sel = np.nonzero(fid == f)[0]
        fr = np.zeros((slots, packet_size), np.uint8)
        fr[mid[sel] // cpp] = pk[sel]
        frames.append(fr)
    if not frames:
        raise RuntimeError("no lidar packets of this size in the capture")
    return np.stack(frames)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--capture", default=CAPTURE)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ouster_sdk_amd import parallel
    from ouster_sdk_amd.device import HotPath

    one_device = os.environ.get("BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(0 if one_device else local_rank)
    if one_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, _, world = parallel.init_from_env(backend)
    dev = torch.device("cuda", torch.cuda.current_device())
    xdev = dev if backend == "nccl" else torch.device("cpu")   # where the exchanged tensors live

    m = read_sensor_json(args.capture + ".json")
    hp = HotPath(m["profile"], m["h"], m["w"], m["cpp"])
    hp.set_pixel_shift_by_row(m["shifts"])
    hp.add_lut(m["b2l"], m["l2s"], m["az"], m["alt"])
    slots = m["w"] // m["cpp"]
    N = args.frames
    frame_shape = (slots, hp.packet_size)

    batch = None
    n_recorded = 0
    if rank == 0:
        rec = recorded_frames(args.capture + ".pcap", hp.packet_size, slots, m["cpp"])
        n_recorded = rec.shape[0]
        idx = np.arange(N) % n_recorded
        batch = torch.from_numpy(rec[idx]).to(xdev)
    b, e = parallel.shard_range(N, rank, world)
    out = hp.alloc_outputs(e - b, destagger=["RANGE", "REFLECTIVITY"], xyz=["RANGE"])

    def sync():
        torch.cuda.synchronize()
        parallel.barrier()

    t_scatter = t_decode = t_gather = 0.0
    gathered = None
    for rep in range(args.reps + 1):       # first pass warms up allocations / RCCL channels / the tuner
        sync()
        t0 = time.perf_counter()
        if world > 1:
            mine = parallel.scatter_frames(batch, N, frame_shape, torch.uint8, xdev)
        else:
            mine = batch
        mine = mine.to(dev)
        sync()
        t1 = time.perf_counter()
        hp.decode(mine, out)
        sync()
        t2 = time.perf_counter()
        xyz = out["xyz:RANGE"] if backend == "nccl" else out["xyz:RANGE"].cpu()
        gathered = parallel.gather_frames(xyz, N) if world > 1 else xyz
        sync()
        t3 = time.perf_counter()
        if rep:
            t_scatter += t1 - t0
            t_decode += t2 - t1
            t_gather += t3 - t2
    reps = args.reps
    if rank == 0:
        g = gathered.to(dev)
        pts = N * m["h"] * m["w"]
        # rank-count independent checksum: integer sum of the f32 bit patterns, plus range sanity
        cks = int(g.view(torch.int32).to(torch.int64).sum().item())
        nz = int((g != 0).any(dim=-1).sum().item())
        line = {
            "workload": "configs[3]: batch of recorded OS-1-128 frames sharded over the GPUs via "
                        + ("RCCL" if backend == "nccl" else backend) + " scatter / gather",
            "capture": os.path.basename(args.capture) + ".pcap", "recorded_frames": n_recorded,
            "frames": N, "h": m["h"], "w": m["w"], "profile": m["profile"], "n_gpus": world,
            "one_device": one_device, "backend": backend,
            "scatter_packets_ms": round(t_scatter / reps * 1e3, 3),
            "decode_ms": round(t_decode / reps * 1e3, 3),
            "gather_xyz_ms": round(t_gather / reps * 1e3, 3),
            "Mpoints_per_s_kernel_only": round(pts / (t_decode / reps) / 1e6, 1),
            "Mpoints_per_s_with_exchange": round(pts / ((t_scatter + t_decode + t_gather) / reps) / 1e6, 1),
            "scatter_GBps": round(N * slots * hp.packet_size * (world - 1) / max(world, 1) / max(t_scatter / reps, 1e-9) / 1e9, 1),
            "gather_GBps": round(pts * 12 * (world - 1) / max(world, 1) / max(t_gather / reps, 1e-9) / 1e9, 1),
            "xyz_checksum": cks, "nonzero_points": nz,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
