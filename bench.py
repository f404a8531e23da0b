#!/usr/bin/env python3
"""bench.py -- Mpoints/s projected (decode + destagger + cartesian), 128x2048 dual return.

A "step" = one pass of the fused hot path over a resident batch of synthetic frames:
  wire packets (HBM) -> 8 staggered planes + 4 destaggered planes + 2 x XYZ f32 (HBM),
i.e. FrameBatcher + destagger<T> + XYZLutT<float>::operator() of the reference for every
frame of the batch (BASELINE.json configs[2]: OS-2-128 2048x128 RNG15_RFL8_NIR8 dual return).
A "point" is one (pixel, return) with XYZ produced: 524288 per frame.

  python bench.py --gpus N --steps K --warmup W
N > 1, one rank per GPU over RCCL: either an external launcher starts the ranks (the driver's
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: WORLD_SIZE is set and must equal N),
or plain `python bench.py --gpus N` starts them itself through the same torch.distributed.run command line
(`plan_launch`); a box with fewer than N GPUs is an error, not a smaller run.  Frames are independent, so
ranks shard the batch with no data-path collective ("weak": per-GPU work fixed).  Rank 0
prints ONE JSON line with the whole-job rate, the roofline of the dominant kernel
(k_decode, timed with HIP events on its stream) and the CPU baseline (the oracle's
restatement of the reference loops, timed on this box's host cores, rank 0 at N=1 only).

Setup before the W warm-up steps (none of it inside the timed region, all of it reported in the JSON line):
  * the library's kernel-variant tuner settles (first 20 calls of a workload shape);
  * buffer placement, `--placement refine` (default): the frugal search that ouster::sdk::hip::DeviceFrameBatch runs by itself
    when it is constructed (BatchOptions::auto_placement = true, the library's default since round 5): two more copies of
    the output set are allocated back to back (6.6 GB transient for 256 dual-return frames, 0.1 s) and every buffer group (XYZ
    pair, 32-bit planes, destaggered planes, narrow planes) is kept at the fastest of its three locations -- then, since round 6
    (`--placement-scan-tries`, default 10; 0 switches it off), what BatchOptions::placement_thorough adds: ten whole output sets
    drawn 4 GB apart (the set kept so far is a candidate), their memory given back, the group-wise search once more (2 - 9 s,
    ~75 GB transient).  One fresh process in three draws only slow placements for its first allocations (0.645 of the roofline per
    step where the same GPU gives 0.70 - 0.75) and the back-to-back draws never leave them; the second group-wise search, in the
    memory the whole sets gave back, does (profiles/r06_latency/placement_notes.txt).  "placement" reports every
    draw and roofline.first_allocation_* the first allocation's time next to the kept one.  `--placement first` takes the first
    allocation as it comes (auto_placement = false) and reports the search beside it (roofline.searched_placement_*);
    `--placement draws` is the round-2 diagnostic;
  * the host-API rows ("drop_in": child processes) are taken before this process owns anything on the GPU;
  * after the K timed steps the outputs are compared with the oracle ("validated", "max_abs_dxyz_m").
The timed steps rotate over `--rotate-inputs` copies of the packet batch (default 2: no step finds its input in the
256 MB Infinity Cache).  After the timed region the paths the metric never touches are timed on the same buffers and
reported under "loss_paths" (never as `value`): `holes` (one lost packet per frame, left as a zeroed slot: optimistic
pass), `stray10` (10 % of the frames compacted after a drop / shuffled: fix-up pass), `general` (every frame compacted
into 127 slots with per-frame packet counts: the general mapping for every frame).
`--workload`, `--outputs`, `--exchange`, `--pcie` select other configurations / ablations.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, CPP = 128, 2048, 16
PROFILE = "RNG15_RFL8_NIR8_DUAL"
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

# wire layout of the benchmark profile (bit_start, bit_size, upshift) -- parsing.cpp:304-315
DUAL_LB_BITS = {"RANGE": (0, 15, 3), "FLAGS": (15, 1, 0), "REFLECTIVITY": (16, 8, 0),
                "NEAR_IR": (24, 8, 4), "RANGE2": (32, 15, 3), "FLAGS2": (47, 1, 0),
                "REFLECTIVITY2": (48, 8, 0), "WINDOW": (56, 8, 0)}
DESTAGGERED = ["RANGE", "RANGE2", "REFLECTIVITY", "REFLECTIVITY2"]
# configs[1]: RNG19_RFL8_SIG16_NIR16 single return, 12 B/px -- parsing.cpp:251-261
SINGLE_BITS = {"RANGE": (0, 19, 0), "FLAGS": (19, 5, 0), "REFLECTIVITY": (32, 8, 0),
               "SIGNAL": (48, 16, 0), "NEAR_IR": (64, 16, 0), "WINDOW": (88, 8, 0)}
WORKLOADS = {
    # name: (profile, bits, chan bytes, destaggered planes, xyz planes, plane B/px, dst B/px, config label)
    "dual": ("RNG15_RFL8_NIR8_DUAL", DUAL_LB_BITS, 8, DESTAGGERED, ["RANGE", "RANGE2"], 15, 10,
             "configs[2]: OS-2-128 2048x128 RNG15_RFL8_NIR8_DUAL dual return"),
    "single": ("RNG19_RFL8_SIG16_NIR16", SINGLE_BITS, 12, ["RANGE", "REFLECTIVITY"], ["RANGE"], 11, 5,
               "configs[1]: OS-1-128 2048x128 RNG19_RFL8_SIG16_NIR16 single return"),
    "fused4": ("RNG15_RFL8_NIR8_DUAL", DUAL_LB_BITS, 8, DESTAGGERED, ["RANGE", "RANGE2"], 15, 10,
               "configs[4] per-GPU share: 4 sensors x dual return per tick, per-sensor extrinsics in-kernel"),
    # configs[3]: a fixed batch of 512 single-return frames split over the ranks (strong scaling)
    "batch512": ("RNG19_RFL8_SIG16_NIR16", SINGLE_BITS, 12, ["RANGE", "REFLECTIVITY"], ["RANGE"], 11, 5,
                 "configs[3]: 512 OS-1-128 2048x128 RNG19_RFL8_SIG16_NIR16 frames sharded over the GPUs"),
}


def synth_calibration():
    """OS-2-128 style calibration."""
    alt = np.linspace(21.0, -21.0, H)
    az = np.tile(np.array([4.2, 1.4, -1.4, -4.2]), H // 4)
    shifts = np.round(az / 360.0 * W).astype(np.int32)
    b2l = np.eye(4)
    b2l[0, 3] = 13.762
    l2s = np.array([[-1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 1, 36.18], [0, 0, 0, 1]], dtype=np.float64)
    return alt, az, shifts, b2l, l2s


def synth_packets(n_frames: int, seed: int = 0xDEADBEEF, zero_frac: float = 0.3,
                  bits=None, chan: int = 8) -> np.ndarray:
    """Wire packets of n_frames random dual-return frames, [n, 128, 16640] uint8.
    Numpy bit packing of the RNG15_RFL8_NIR8_DUAL layout with STANDARD headers
    (packet header: type u16 | frame_id u16 | init_id u24 | prod_sn u40; column header:
    timestamp u64 | measurement_id u16 | status u16) -- independent of oracle/ and of the
    library under test."""
    bits = bits or DUAL_LB_BITS
    col_size = 12 + H * chan
    pkt_size = 32 + CPP * col_size + 32
    ppf = W // CPP
    out = np.zeros((n_frames, ppf, pkt_size), dtype=np.uint8)
    for f in range(n_frames):
        rng = np.random.default_rng(seed + f)
        lo = np.zeros((W, H), dtype=np.uint64)  # pixel bits 0..63, column major like the wire
        hi = np.zeros((W, H), dtype=np.uint64)  # pixel bits 64..127
        for name, (start, size, up) in bits.items():
            v = rng.integers(0, 1 << size, size=(W, H), dtype=np.uint64)
            if name in ("RANGE", "RANGE2"):
                v[rng.random((W, H)) < zero_frac] = 0
            if start < 64:
                lo |= v << np.uint64(start)
            else:
                hi |= v << np.uint64(start - 64)
        raw = np.concatenate([lo.view(np.uint8).reshape(W, H, 8), hi.view(np.uint8).reshape(W, H, 8)],
                             axis=2)[:, :, :chan]
        cols = np.zeros((W, col_size), dtype=np.uint8)
        cols[:, 0:8] = (1000 + np.arange(W, dtype=np.uint64)).view(np.uint8).reshape(W, 8)
        cols[:, 8:10] = np.arange(W, dtype=np.uint16).view(np.uint8).reshape(W, 2)
        cols[:, 10] = 1
        cols[:, 12:] = raw.reshape(W, H * chan)
        out[f, :, 32:32 + CPP * col_size] = cols.reshape(ppf, CPP * col_size)
        hdr = np.zeros(32, dtype=np.uint8)
        hdr[0] = 1
        hdr[2:4] = np.frombuffer(np.uint16((700 + f) & 0xFFFF).tobytes(), np.uint8)
        hdr[4:7] = [0x56, 0x34, 0x12]
        hdr[7:12] = [0x55, 0x44, 0x33, 0x22, 0x11]
        out[f, :, :32] = hdr
    return out


def algorithmic_bytes_per_frame(workload: str = "dual") -> int:
    """packets + planes + destaggered planes + XYZ f32 (separable LUT tables:
    no LUT bytes).  dual (config 3) = 14 974 976 B, single (config 2) = 10 518 528 B."""
    _, _, chan, _, xyz_names, plane_b, dst_b, _ = WORKLOADS[workload]
    pkts = (W // CPP) * (32 + CPP * (12 + H * chan) + 32)
    return pkts + H * W * plane_b + H * W * dst_b + len(xyz_names) * H * W * 3 * 4


def usable_cores():
    """Host cores this process may actually keep busy: the affinity mask capped by the cgroup CPU quota.  (os.cpu_count() says
    256 on the GPU boxes; round 6's first all-core run got 3 x one core out of 256 spinning threads and 2 Mpoints/s out of the
    reference's OpenMP cartesian -- a quota a fraction of the machine wide throttles a team that size.)  Returns
    (threads to use, {how it was derived})."""
    n_aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    info = {"cpu_count": os.cpu_count(), "affinity": n_aff, "cgroup_quota_cores": None}
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:            # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:   # cgroup v1
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    n = n_aff
    if quota is not None:
        info["cgroup_quota_cores"] = round(quota, 2)
        n = max(1, min(n_aff, int(quota + 0.5)))
    return n, info


def cpu_baseline(pool: np.ndarray, shifts, target_s: float = 8.0):
    """The reference's CPU path on a bounded sample, timed on this box's host cores.

    value (kind "reference"): the REFERENCE's own loops compiled from its sources (oracle/_ref, oracle/Makefile) -- block_field of
    every plane for every packet (FrameBatcher's block path), destagger_into x4, cartesianT<double> x2 -- over the same 16-frame
    pool the port iterates, one thread, as the reference ships (oracle/hotpath_ref.cpp is the harness; it holds no reference
    code).  `port` beside it: the oracle's C restatement of the whole FrameBatcher path (state machine, headers, zero fill
    included).  all_cores: the same two with the pool's frames spread over every host core, plus cartesianT built with
    -DOUSTER_OMP, the reference's own parallel form.  Without oracle/_ref (a tree that never saw /root/reference) the port is
    `value` and kind says so."""
    from oracle import oracle as O
    import ctypes as C
    O.build()
    cal = O.synthetic_calib(h=H, w=W, profile=PROFILE)
    pf = cal.packet_format()
    ldir, lofs = cal.xyz_lut(False)
    n = pool.shape[0]
    flat = np.ascontiguousarray(pool)
    sh = np.ascontiguousarray(shifts, dtype=np.int32)
    cks = C.c_uint64()
    cores, core_info = usable_cores()
    pts_per_frame = H * W * 2
    bpp = algorithmic_bytes_per_frame("dual") / pts_per_frame

    def run(reps, threads, f64, n_virtual=None):
        return O.lib().ora_bench_hot_path(C.byref(pf), 1, flat.ctypes.data, n, n_virtual or n,
                                          W // CPP, sh.ctypes.data, ldir.ctypes.data,
                                          lofs.ctypes.data, int(f64), reps, threads, C.byref(cks))

    # ---- the oracle's port, one core ----------------------------------------------------------------------------------
    t1 = run(1, 1, True)                                   # calibrate
    reps = max(1, int(0.5 * target_s / max(t1, 1e-3)))
    t = run(reps, 1, True)
    port = {"value": n * reps * pts_per_frame / t / 1e6, "unit": "Mpoints/s", "cores": 1, "kind": "port",
            "sample": f"{n} frames x {reps} passes, FrameBatcher(block path, state machine, headers)+destagger x4+"
                      f"cartesianT<double> x2, {t:.1f} s"}
    rf = max(1, reps // 3)
    tf = run(rf, 1, False)
    port["f32_variant"] = {"value": n * rf * pts_per_frame / tf / 1e6, "cores": 1}

    # ---- the reference's own loops, one core (the baseline `value`) -----------------------------------------------------------
    ref, hp = None, None
    try:
        from oracle import hotpath_ref
        if hotpath_ref.available():
            fr = O.Frame.for_profile(cal.profile, H, W, CPP, with_window=True)
            dtypes = {nm: fr.plane(nm).dtype for nm in fr.plane_names()}
            hp = hotpath_ref.HotPath(O, pf, pool, dtypes, DESTAGGERED, ["RANGE", "RANGE2"], ldir, lofs, sh)
            tw, _, _, _ = hp.run(n, 1)
            rr = max(1, int(target_s / max(tw, 1e-3)))
            tw, legs, planes, cloud = hp.run(n, rr, want_outputs=True)
            # thread 0's last frame is frame n - 1 of the pool: what the oracle decodes from the same packets
            O.batch_frame(pf, pool[n - 1], fr, init_id=O.lib().ora_init_id(C.byref(pf), pool[n - 1][0].ctypes.data))
            same = all(np.array_equal(planes[nm], fr.plane(nm)) for nm in planes)
            same = same and bool(np.array_equal(cloud, O.cartesian(fr.plane("RANGE"), ldir, lofs)))
            per = n * rr
            ref = {"value": per * pts_per_frame / tw / 1e6, "unit": "Mpoints/s", "cores": 1, "kind": "reference",
                   "sample": f"{n} frames x {rr} passes: block_field<T,16> of {len(hp.names)} planes x {W // CPP} packets (parsing.cpp:628-657) + "
                             f"destagger_into x{len(DESTAGGERED)} (lidar_frame_impl.h:733-760) + cartesianT<double> x2 (impl/cartesian.h:36-66), the "
                             f"reference's code compiled from its sources (oracle/_ref), {tw:.1f} s",
                   "ms_per_frame": {"decode": round(legs[0] / per * 1e3, 3), "destagger": round(legs[1] / per * 1e3, 3),
                                    "cartesian": round(legs[2] / per * 1e3, 3), "total": round(tw / per * 1e3, 3)},
                   "equals_oracle": bool(same),
                   "not_included": "FrameBatcher's per-packet bookkeeping and the 28 KB of column headers per frame (the port has them)"}
    except Exception as e:   # noqa: BLE001 -- reported; the port then stands in
        ref = {"error": f"{type(e).__name__}: {e}"[:200]}

    if ref and "value" in ref:
        res = dict(ref)
        res["port"] = port
        res["port_over_reference"] = round(port["value"] / ref["value"], 3)
    else:
        res = dict(port)
        if ref:
            res["reference_error"] = ref.get("error")
    res["f32_variant"] = port["f32_variant"]

    # ---- every host core: frames of the pool over threads ---------------------------------------------------------------------------
    if cores > 1:
        threads = cores
        nv = threads * 4

        def run_all(reps_, flags):
            return O.lib().ora_bench_hot_path2(C.byref(pf), 1, flat.ctypes.data, n, nv, W // CPP, sh.ctypes.data,
                                               ldir.ctypes.data, lofs.ctypes.data, 1, reps_, threads, C.byref(cks), flags)
        # flags 3: every thread on its own first-touched copy of the LUT / packets (NUMA-local pages), static schedule
        ta = run_all(1, 3)
        ra = max(1, int(3.0 / max(ta, 1e-3)))
        ta = run_all(ra, 3)
        cp_bytes, cp_reps = 64 << 20, 8
        tcopy = O.lib().ora_bench_stream_copy(cp_bytes, cp_reps, threads)
        allc = {"cores": threads, "host_cores": core_info,
                "port_value": round(nv * ra * pts_per_frame / ta / 1e6, 1),
                "stream_copy_GBps_same_threads": round(2.0 * cp_bytes * cp_reps * threads / tcopy / 1e9, 1),
                "note": "GBps counts SURVEY 8(d)'s algorithmic bytes; the f64 path also reads 2 x 24 B of LUT per point, so its "
                        "memory traffic is about 3.5 x that"}
        if hp is not None:
            try:
                from oracle import hotpath_ref
                tr = hp.run(nv, 1, threads=threads, own_inputs=True)[0]
                rra = max(1, int(4.0 / max(tr, 1e-3)))
                tr = hp.run(nv, rra, threads=threads, own_inputs=True)[0]
                allc.update({"value": nv * rra * pts_per_frame / tr / 1e6, "kind": "reference",
                             "sample": f"{nv} frames x {rra} passes, the frames spread over {threads} OpenMP threads (static schedule), every "
                                       "thread on its own first-touched planes, cloud, LUT and packet copies; the reference's loops "
                                       "(oracle/_ref) as in `value`"})
                if hotpath_ref.omp_available():
                    fr0 = O.Frame.for_profile(cal.profile, H, W, CPP, with_window=True)
                    O.batch_frame(pf, pool[0], fr0, init_id=O.lib().ora_init_id(C.byref(pf), pool[0][0].ctypes.data))
                    to = hotpath_ref.bench_cartesian_omp(fr0.plane("RANGE"), ldir, lofs, 4, threads)
                    ro = max(4, int(1.5 / max(to / 4, 1e-5)))
                    to = hotpath_ref.bench_cartesian_omp(fr0.plane("RANGE"), ldir, lofs, ro, threads)
                    t1c = hotpath_ref.bench_cartesian_omp(fr0.plane("RANGE"), ldir, lofs, 8, 1)
                    allc["reference_omp_cartesian"] = {
                        "Mpoints_per_s": round(H * W * ro / to / 1e6, 1), "threads": threads,
                        "one_thread_Mpoints_per_s": round(H * W * 8 / t1c / 1e6, 1),
                        "what": "cartesianT<double> built with -fopenmp -DOUSTER_OMP (impl/cartesian.h:15-23,50-52): the reference's own "
                                "parallel form, `omp parallel for schedule(static)` over the points of ONE cloud (one frame at a time)"}
            except Exception as e:   # noqa: BLE001
                allc["reference_error"] = f"{type(e).__name__}: {e}"[:200]
        if "value" not in allc:
            allc.update({"value": allc["port_value"], "kind": "port",
                         "sample": f"{nv} frames x {ra} passes over {threads} OpenMP threads, per-thread first-touched LUT and packet "
                                   "copies, static schedule"})
        res["all_cores"] = allc
    # the same algorithmic byte count as the GPU leg: bytes/s next to points/s
    res["GBps"] = res["value"] * 1e6 * bpp / 1e9
    if "all_cores" in res:
        res["all_cores"]["GBps"] = res["all_cores"]["value"] * 1e6 * bpp / 1e9
    return res


def validate_against_oracle(hp, profile, packets, out, shifts, lut_args, n_luts, sample):
    """Outside the timed region: decode the sampled frames of the batch with the CPU oracle (the
    restatement of the reference loops) and compare every plane, destaggered plane, column header
    and XYZ cloud the timed steps produced.  Returns (ok, max |dXYZ| in metres, frames checked)."""
    import ctypes as C
    from oracle import oracle as O
    O.build()
    cal = O.synthetic_calib(h=H, w=W, profile=profile)
    pf = cal.packet_format()
    luts = []
    for (b2l, tf, az, alt) in lut_args:
        luts.append(O.make_xyz_lut(W, H, 0.001, b2l, tf, az, alt))
    names = [n for n, _ in hp.fields]
    ok, worst = True, 0.0
    for f in sample:
        pk = packets[f].cpu().numpy()
        fr = O.Frame.for_profile(cal.profile, H, W, CPP, with_window=True)
        fr.fill(0xAB)
        O.batch_frame(pf, pk, fr, init_id=O.lib().ora_init_id(C.byref(pf), pk[0].ctypes.data))
        for n in names:
            if n in out:
                ok &= bool(np.array_equal(out[n][f].cpu().numpy(), fr.plane(n)))
            if "destaggered:" + n in out:
                ok &= bool(np.array_equal(out["destaggered:" + n][f].cpu().numpy(),
                                          O.destagger(fr.plane(n), shifts)))
        for k, ref in (("timestamp", fr.timestamp), ("measurement_id", fr.measurement_id), ("status", fr.status)):
            if k in out:
                ok &= bool(np.array_equal(out[k][f].cpu().numpy(), ref))
        d, o = luts[f % n_luts]
        for n in names:
            if "xyz:" + n in out:
                want = O.cartesian(fr.plane(n), d, o)
                got = out["xyz:" + n][f].cpu().numpy().astype(np.float64)
                worst = max(worst, float(np.abs(got - want).max()))
                ok &= bool(np.all(got[fr.plane(n).reshape(-1) == 0] == 0))
    ok &= worst <= 1e-4
    return ok, worst, list(sample)


def kernel_sources_sha256() -> str:
    """Identity of the kernel sources a committed PMC figure was recorded on (the GPU box has no .git)."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "ouster_sdk_amd", "csrc")
    for name in ("kernels_common.h", "wide_tile.h", "k_decode.hip", "k_decode_stream.hip", "ouster_hip_dev.h", "ouster_hip_capi.hip"):
        with open(os.path.join(base, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def time_loss_paths(hp, packets, out, F, bytes_per_frame, steps=30):
    """Decode rates of the paths the metric never touches, on the buffers of the timed run:
      holes    one packet of every frame lost and left as a zeroed slot (how FrameStream / DeviceFrameBatch stage a
               lossy stream): the optimistic pass alone;
      stray10  10 % of the frames with live columns outside their home slots (compacted after a drop, or two packets
               swapped): flagged by the optimistic pass, redone by the persistent fix-up pass;
      general  every frame compacted into W/cpp - 1 slots + per-frame packet counts: slots * cpp != W, the general
               mapping (MODE_GENERAL) for every frame -- the reference's parse_by_col fallback, lidar_frame.cpp:1422-1466.
    Algorithmic bytes: the packets that are present + all outputs (missing columns are written as zeros)."""
    import torch
    slots, stride = packets.shape[1], packets.shape[2]
    res = {}
    f_idx = torch.arange(F, device="cuda")
    lost = (f_idx * 7 + 3) % slots                       # the packet every frame loses
    keep = torch.arange(slots, device="cuda").unsqueeze(0).expand(F, slots)
    keep = keep[keep != lost.unsqueeze(1)].reshape(F, slots - 1)      # [F, slots-1] surviving packet indices, in order

    def clock(pk, counts):
        for _ in range(24):                               # variant tuner of this shape + warm-up
            hp.decode(pk, out, packet_counts=counts)
        torch.cuda.synchronize()
        hp.ctx.timing(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            hp.decode(pk, out, packet_counts=counts)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        kms, _ = hp.ctx.timing_read()
        hp.ctx.timing(False)
        return dt, kms

    def report(name, dt, kms, present_packets, what):
        nbytes = bytes_per_frame * F - (F * slots - present_packets) * (stride)
        tc, tr = hp.ctx.last_decode_tile()
        res[name] = {"what": what, "ms_per_step": round(dt * 1e3, 4), "first_pass_kernel_ms": round(kms, 4),
                     "kernel": f"{hp.ctx.last_decode_kernel()} {tc}x{tr}",
                     "frac": round(nbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if kms > 0 else None,
                     "frac_step": round(nbytes / dt / 1e9 / HBM_PEAK_GBPS, 4),
                     "algorithmic_bytes_per_launch": int(nbytes)}

    # holes
    pk = packets.clone()
    pk[f_idx, lost] = 0
    dt, kms = clock(pk, None)
    report("holes", dt, kms, F * (slots - 1), "one packet per frame lost, left as a zeroed slot: optimistic pass only")
    # stray10: every 20th frame compacted after its drop (count slots-1), every 20th + 10 has two packets swapped
    pk = packets.clone()
    counts = torch.full((F,), slots, dtype=torch.int32, device="cuda")
    comp = f_idx[f_idx % 20 == 3]
    if comp.numel():
        pk[comp, :slots - 1] = packets[comp.unsqueeze(1), keep[comp]]
        pk[comp, slots - 1] = 0
        counts[comp] = slots - 1
    swp = f_idx[f_idx % 20 == 13]
    if swp.numel():
        a, b = packets[swp, 10].clone(), packets[swp, 11].clone()
        pk[swp, 10], pk[swp, 11] = b, a
    dt, kms = clock(pk, counts)
    n_fix = int(comp.numel() + swp.numel())
    report("stray10", dt, kms, F * slots - int(comp.numel()),
           f"{n_fix} of {F} frames with strays (half compacted after a drop, half with two packets swapped): "
           "optimistic pass + fix-up pass over those frames")
    # general: [F, slots-1] buffer, every frame compacted
    pk = packets[f_idx.unsqueeze(1), keep].contiguous()
    counts = torch.full((F,), slots - 1, dtype=torch.int32, device="cuda")
    dt, kms = clock(pk, counts)
    report("general", dt, kms, F * (slots - 1),
           f"every frame compacted into {slots - 1} slots with per-frame packet counts: general mapping for every frame")
    return res


def plan_launch(gpus: int, environ, argv, n_devices=None):
    """What `bench.py --gpus N` does about its ranks (no torch import: unit-tested on CPU).
      ("run", world)   this process is one rank of `world` (1 without a launcher); under an external launcher
                       (WORLD_SIZE set) --gpus must agree with it, anything else is a mis-launch and raises;
      ("spawn", cmd)   no launcher and N > 1: the torch.distributed.run command line that starts N ranks of this script,
                       one per GPU, rendezvous on 127.0.0.1 at a free port.
    `n_devices` (when known) must cover N unless BENCH_ONE_DEVICE=1 (test knob: every rank on cuda:0)."""
    if gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" in environ:
        world = int(environ["WORLD_SIZE"])
        if world != gpus:
            raise SystemExit(f"bench.py --gpus {gpus} was started by a launcher with WORLD_SIZE={world}: "
                             "the two must agree (n_gpus in the JSON line is the number of ranks that ran)")
        return "run", world
    if gpus == 1:
        return "run", 1
    if n_devices is not None and n_devices < gpus and environ.get("BENCH_ONE_DEVICE") != "1":
        raise SystemExit(f"bench.py --gpus {gpus}: only {n_devices} GPU(s) visible on this box")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    return "spawn", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
                     "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def _workload_setup(name, n_frames, pool_frames=4, seed=0xC0FFEE):
    """HotPath + LUT(s) + a resident packet batch + outputs of one BASELINE config (first allocation, no placement search)."""
    import torch
    from ouster_sdk_amd.device import HotPath
    profile, bits, chan, dst_names, xyz_names, _, _, label = WORKLOADS[name]
    alt, az, shifts, b2l, l2s = synth_calibration()
    hp = HotPath(profile, H, W, CPP)
    hp.set_pixel_shift_by_row(shifts)
    lut_args = []
    if name == "fused4":
        for k in range(4):
            a = 0.5 * np.pi * k
            ext = np.array([[np.cos(a), -np.sin(a), 0, 1000.0 * k], [np.sin(a), np.cos(a), 0, -500.0 * k],
                            [0, 0, 1, 250.0], [0, 0, 0, 1]])
            lut_args.append((b2l, ext @ l2s, az, alt))
    else:
        lut_args.append((b2l, l2s, az, alt))
    for la in lut_args:
        hp.add_lut(*la)
    pool = torch.from_numpy(synth_packets(pool_frames, seed=seed, bits=bits, chan=chan)).cuda()
    packets = pool.repeat((n_frames + pool_frames - 1) // pool_frames, 1, 1)[:n_frames].contiguous()
    out = hp.alloc_outputs(n_frames, destagger=dst_names, xyz=xyz_names)
    return hp, packets, out, profile, shifts, lut_args, len(xyz_names), label


def time_other_workloads(steps=12, placement="refine", scan_tries=10, scan_stride_gb=4.0):
    """The BASELINE configs the metric is not quoted on, each timed like the headline (tuner settled, the same frugal
    placement search -- two more output sets back to back -- unless `placement` is "first", two input batches in turn, HIP
    events around the decode kernel) and checked against the oracle -- report rows, never `value`:
    single = configs[1], batch512 = configs[3] (this GPU's share at N = 1: all 512 frames), fused4 = configs[4]'s per-GPU
    share (64 ticks x 4 sensors, per-sensor extrinsics in the kernel)."""
    import torch
    res = {}
    for name, F in (("single", 256), ("batch512", 512), ("fused4", 256)):
        hp, packets, out, profile, shifts, lut_args, n_ret, label = _workload_setup(name, F)
        for _ in range(24):
            hp.decode(packets, out)
        torch.cuda.synchronize()
        first_ms, scanned = None, 0
        if placement == "refine":
            out, rep = hp.refine_placement(packets, out, draws=3, ballast_gb=0.0)
            first_ms = rep["first_allocation_ms"]
            if scan_tries >= 2:   # the headline's sequence: whole sets drawn, then the group-wise search again in the memory they gave back
                prof_, bits_, chan_, dst_, xyz_ = WORKLOADS[name][:5]
                out_bytes = sum(v.numel() * v.element_size() for v in out.values())
                free_b, _ = torch.cuda.mem_get_info()
                tries = min(scan_tries, int(free_b * 0.6) // (out_bytes + int(scan_stride_gb * (1 << 30))) - 1)
                if tries >= 2:
                    try:
                        packets, out, srep = hp.pick_placement(packets, lambda: hp.alloc_outputs(F, destagger=dst_, xyz=xyz_), tries=tries,
                                                               launches=8, stride_gb=scan_stride_gb, incumbent=out, slab=False)
                        out, rep = hp.refine_placement(packets, out, draws=3, ballast_gb=0.0)
                        scanned = len(srep["output_sets_ms"])
                    except Exception:
                        torch.cuda.empty_cache()
        inputs = [packets, packets.clone()]
        hp.ctx.timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            hp.decode(inputs[i & 1], out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        kms, _ = hp.ctx.timing_read()
        hp.ctx.timing(False)
        tc, tr = hp.ctx.last_decode_tile()
        nbytes = algorithmic_bytes_per_frame(name) * F
        ok, worst, checked = validate_against_oracle(hp, profile, packets, out, shifts, lut_args, len(lut_args),
                                                     sorted({0, 5 % F, F - 1}))
        res[name] = {"config": label, "frames_per_step": F, "kernel": f"{hp.ctx.last_decode_kernel()} {tc}x{tr}",
                     "kernel_ms": round(kms, 4), "ms_per_step": round(dt * 1e3, 4),
                     "Mpoints_per_s": round(F * H * W * n_ret / dt / 1e6, 1),
                     "frac": round(nbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if kms > 0 else None,
                     "frac_step": round(nbytes / dt / 1e9 / HBM_PEAK_GBPS, 4),
                     "algorithmic_bytes_per_launch": int(nbytes), "validated": bool(ok), "max_abs_dxyz_m": worst,
                     "validated_frames": checked,
                     "buffer_placement": "first allocation" if first_ms is None else
                                         ("fastest of 3 back-to-back locations per buffer group" +
                                          (f", then of that and {scanned} further whole output sets, then per group again (like the headline)" if scanned else "")),
                     "first_allocation_ms_per_call": first_ms}
        del hp, packets, out, inputs
        torch.cuda.empty_cache()
    return res


def time_small_batches(calls=300):
    """Latency view (the single-frame number is reported separately): ouster_hip_decode on 1, 4 and 16
    dual-return frames with the full output set.  4 frames with four LUTs = ONE configs[4] tick (4 sensors x 128 x 2048 dual
    return, per-sensor extrinsics in the kernel).  `us_per_call_pipelined`: calls issued back to back on the stream (a
    streaming caller); `us_per_call_sync`: call + stream synchronisation every time (median; what one tick waits for)."""
    import torch
    res = {}
    for n in (1, 4, 16):
        hp, packets, out, profile, shifts, lut_args, n_ret, _ = _workload_setup("fused4" if n == 4 else "dual", n)
        inputs = [packets, packets.clone()]
        for _ in range(24):
            hp.decode(packets, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(calls):
            hp.decode(inputs[i & 1], out)
        torch.cuda.synchronize()
        pipelined = (time.perf_counter() - t0) / calls
        lat = []
        for i in range(60):
            t0 = time.perf_counter()
            hp.decode(inputs[i & 1], out)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        hp.ctx.timing(True)
        for i in range(20):
            hp.decode(inputs[i & 1], out)
        torch.cuda.synchronize()
        kms, _ = hp.ctx.timing_read()
        hp.ctx.timing(False)
        tc, tr = hp.ctx.last_decode_tile()
        nbytes = algorithmic_bytes_per_frame("dual") * n
        ok, worst, _ = validate_against_oracle(hp, profile, packets, out, shifts, lut_args, len(lut_args), list(range(min(n, 4))))
        res[str(n)] = {"frames": n, "what": "one configs[4] tick: 4 sensors, per-sensor extrinsics" if n == 4 else f"{n} dual-return frame(s)",
                       "kernel": f"{hp.ctx.last_decode_kernel()} {tc}x{tr}", "first_pass_kernel_us": round(kms * 1e3, 2),
                       "us_per_call_pipelined": round(pipelined * 1e6, 2), "us_per_call_sync": round(float(np.median(lat)) * 1e6, 2),
                       "Gpoints_per_s_pipelined": round(n * H * W * n_ret / pipelined / 1e9, 2),
                       "frac_pipelined": round(nbytes / pipelined / 1e9 / HBM_PEAK_GBPS, 4),
                       "frac_kernel": round(nbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if kms > 0 else None,
                       "validated": bool(ok), "max_abs_dxyz_m": worst}
        del hp, packets, out, inputs
        torch.cuda.empty_cache()
    return res


def time_standalone(n_images=256):
    """The standalone kernels behind ouster_hip_destagger / _cartesian / _dewarp_frames_rows (SURVEY 8(a) rows a10, a13, f-2;
    the reference's own three benchmarks: tests/benchmarks/core_benchmark.cpp:29-154) on resident data, each checked against
    the oracle on one image, `frac` on SURVEY 8(d)'s per-kernel byte counts -- report rows, never `value`.  Every input
    exists in 3 copies used in turn (a call must not find its input in the 256 MB Infinity Cache)."""
    import torch
    from ouster_sdk_amd.device import HotPath
    from oracle import oracle as O
    O.build()
    N = n_images
    alt, az, shifts, b2l, l2s = synth_calibration()
    hp = HotPath(PROFILE, H, W, CPP)
    hp.set_pixel_shift_by_row(shifts)
    lut = hp.add_lut(b2l, l2s, az, alt)
    d64, o64 = lut.export(W, H)
    lut32 = hp.add_lut_arrays(d64.astype(np.float32), o64.astype(np.float32))
    g = torch.Generator(device="cuda").manual_seed(5)
    rng = torch.randint(0, 2 ** 19, (N, H, W), dtype=torch.int64, device="cuda", generator=g)
    rng = (rng * (torch.rand(rng.shape, device="cuda", generator=g) >= 0.3)).to(torch.uint32)   # ~30 % no-return pixels

    class Rot:
        def __init__(self, t):
            self.c, self.i = [t, t.clone(), t.clone()], 0

        def __call__(self):
            self.i += 1
            return self.c[self.i % 3]

    def clock(fn, reps=10):
        fn(); fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e-3

    res, npx = {}, N * H * W
    for name, t, kern in (("destagger_u8", rng.to(torch.uint8), "k_destagger"), ("destagger_u16", rng.to(torch.uint16), "k_destagger"),
                          ("destagger_u32", rng, "k_destagger")):
        rt = Rot(t)
        s = clock(lambda: hp.destagger(rt()))
        ok = bool(np.array_equal(hp.destagger(t[:1].contiguous())[0].cpu().numpy(), O.destagger(t[0].cpu().numpy(), shifts)))
        nbytes = 2 * t.numel() * t.element_size()
        res[name] = {"kernel": kern, "ms": round(s * 1e3, 4), "algorithmic_bytes": int(nbytes),
                     "frac": round(nbytes / s / 1e9 / HBM_PEAK_GBPS, 4), "validated": ok}
    for name, l, bpp, ld, lo in (("cartesian_sep_f32", lut, 4 + 12, d64, o64),
                                 ("cartesian_fullLUT_f32", lut32, 4 + 12 + 24.0 / N, d64.astype(np.float32), o64.astype(np.float32))):
        rr = Rot(rng)
        s = clock(lambda: hp.cartesian(rr(), lut=l, dtype=torch.float32))
        got = hp.cartesian(rng[:1].contiguous(), lut=l, dtype=torch.float32)[0].cpu().numpy().astype(np.float64)
        want = O.cartesian(rng[0].cpu().numpy(), ld.astype(np.float64), lo.astype(np.float64))
        err = float(np.abs(got - want).max())
        nbytes = npx * bpp
        res[name] = {"kernel": "k_cartesian_tiled", "ms": round(s * 1e3, 4), "algorithmic_bytes": int(nbytes),
                     "Mpoints_per_s": round(npx / s / 1e6, 1), "frac": round(nbytes / s / 1e9 / HBM_PEAK_GBPS, 4),
                     "validated": bool(err <= 1e-4), "max_abs_dxyz_m": err}
    # range-gated, compacting frame dewarp on the route DeviceFrameBatch::dewarp takes: the decode's gate counts + float
    # pose rows (impl/dewarp_impl.h:23-81); bytes: range plane + 12 B per kept point + status, pose rows and gate counts per column
    pk = torch.from_numpy(synth_packets(8)).cuda().repeat(N // 8, 1, 1).contiguous()
    dout = hp.alloc_outputs(N, planes=["RANGE"], xyz=[])
    hp.decode(pk, dout, gate=(0.5, 400.0))
    drng, dst_, dgc = dout["RANGE"], dout["status"], dout["gate_counts"]
    ang = torch.linspace(-0.2, 0.2, W, dtype=torch.float64, device="cuda")
    poses = torch.eye(4, dtype=torch.float64, device="cuda").repeat(N, W, 1, 1).contiguous()
    poses[..., 0, 0] = torch.cos(ang); poses[..., 0, 1] = -torch.sin(ang)
    poses[..., 1, 0] = torch.sin(ang); poses[..., 1, 1] = torch.cos(ang)
    poses[..., 0, 3] = 3.0
    rows = HotPath.pose_rows(poses)
    o = hp.dewarp_frames(drng, dst_, rows, 0.5, 400.0, provenance=False, luts=[lut], gate_counts=dgc)
    kept = int(o["frame_offsets"][-1].item())
    rr = Rot(drng)
    s = clock(lambda: hp.dewarp_frames(rr(), dst_, rows, 0.5, 400.0, provenance=False, luts=[lut], gate_counts=dgc))
    k1 = int(o["frame_offsets"][1].item())
    want = O.dewarp_frame(drng[0].cpu().numpy(), dst_[0].cpu().numpy(), np.zeros(W, np.uint64), poses[0].cpu().numpy(),
                          d64.astype(np.float32), o64.astype(np.float32), 0.5, 400.0)
    got = o["points"][:k1].cpu().numpy().astype(np.float64)
    ok = len(want[0]) == k1 and (k1 == 0 or float(np.abs(got - want[0].astype(np.float64)).max()) <= 1e-4)
    nbytes = npx * 4 + kept * 12 + N * W * (48 + 4) + N * W * 2 * 8
    res["dewarp_frames_rows_counted"] = {"kernel": "k_dwf_scan + k_dwf_frame_scan + k_dwf_emit", "ms": round(s * 1e3, 4),
                                         "algorithmic_bytes": int(nbytes), "kept_fraction": round(kept / npx, 3),
                                         "Mpixels_per_s": round(npx / s / 1e6, 1),
                                         "frac": round(nbytes / s / 1e9 / HBM_PEAK_GBPS, 4), "validated": bool(ok)}
    res["note"] = f"{N} images of {H}x{W}, inputs rotated over 3 copies; each row checked against the oracle on image 0"
    res["osf_decode_device"] = _report_row(time_osf)
    return res


def time_osf(n_msgs=96):
    """LidarScan messages of the reference's own OSF fixture (tests/golden/osf: PNG-encoded 128 x 1024 fields) ->
    planes in HBM through OsfFrameDecoder.decode_device -- zlib inflate on the host, PNG scanline filters + unpack + stagger on
    the GPU (k_osf_png_unfilter, k_osf_unpack); the first message is checked against the OSF oracle.  Host bound (inflate)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_osf as T
    from oracle import osf_oracle as Z
    from ouster_sdk_amd import core
    path = T.LB
    meta = list(Z.OsfFile(path).sensor_metadata().values())[0]
    h, w, shifts = T._geometry(meta)
    pf = core.OsfFile(path)
    streams = pf.lidar_scan_streams()
    msgs = [m for (_, sid, m) in pf.messages() if sid in streams]
    batch = (msgs * (n_msgs // len(msgs) + 1))[:n_msgs]
    info = T._sensor_info(core, meta)
    out = {"messages": n_msgs, "h_w": [h, w], "fixture": os.path.relpath(path, ROOT)}
    for on in (True, False):
        dec = core.OsfFrameDecoder(info)
        dec.device_unfilter = on
        dec.decode_device(batch[:8])
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            dec.decode_device(batch)
            best = min(best, time.perf_counter() - t0)
        out["ms_gpu_unfilter" if on else "ms_host_unfilter"] = round(best * 1e3, 2)
    fr = core.OsfFrameDecoder(info).decode([msgs[0]])[0]
    want = Z.decode_lidar_scan_msg(msgs[0], h, w, shifts)
    out["validated"] = bool(all(np.array_equal(fr.field(n), v) for n, v in want["fields"].items()))
    out["frames_per_s"] = round(n_msgs / (out["ms_gpu_unfilter"] * 1e-3), 1)
    return out


def time_latency_cpp(python_rows):
    """The small-batch latency a C++ caller of the library sees (the product is a C++ library; Python is test plumbing):
    tests/cpp/_build/bench_latency (tools/bench_latency.cpp, built by `make`) times hip::DeviceFrameBatch::decode() on 1, 4
    (four sensors = one configs[4] tick) and 16 dual-return frames, full output set -- back to back, with a stream
    synchronisation per call (median), and the host time of the call itself.  The same shapes are decoded and checked against
    the oracle from Python (`latency_python`, whose `validated` flags are copied here)."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "bench_latency")
    o = subprocess.run([exe, "1500"], capture_output=True, text=True, timeout=120)
    if o.returncode != 0:
        raise RuntimeError((o.stderr or o.stdout)[-200:])
    rows = json.loads(o.stdout.strip().splitlines()[-1])
    out = {"timed_by": "tests/cpp/_build/bench_latency (tools/bench_latency.cpp): hip::DeviceFrameBatch::decode() from C++; "
                       "the Python-timed figures of the same calls are under latency_python"}
    for k in ("1", "4", "16"):
        r = dict(rows[k])
        r["what"] = "one configs[4] tick: 4 sensors, per-sensor extrinsics" if k == "4" else f"{k} dual-return frame(s)"
        r["Gpoints_per_s_pipelined"] = round(int(k) * H * W * 2 / (r["us_per_call_pipelined"] * 1e-6) / 1e9, 2)
        if isinstance(python_rows, dict) and k in python_rows:
            r["validated"] = python_rows[k].get("validated")
            r["kernel"] = python_rows[k].get("kernel")
        out[k] = r
    return out


def time_drop_in():
    """What an UNMODIFIED caller of the reference's API gets through include/ouster/core/*.h (host containers in, host results
    out, every call crosses PCIe): FrameBatcher::batch x128 + destagger<uint32_t> + XYZLut() per 128 x 2048 dual-return frame
    (tools/bench_host_api.cpp), and FrameStream host-to-host (tools/bench_stream.cpp: pinned staging, H2D / decode / D2H
    overlapped).  Both are small C++ programs built by `make` against libouster_core_amd.so; never `value`."""
    import subprocess
    res = {}
    bdir = os.path.join(ROOT, "tests", "cpp", "_build")
    for key, argv in (("host_api", ["bench_host_api", "40"]), ("frame_stream", ["bench_stream", "768", "32", "3", "xyz"]),
                      ("frame_stream_compact", ["bench_stream", "1536", "16", "4", "compact"])):
        exe = os.path.join(bdir, argv[0])
        try:
            o = subprocess.run([exe] + argv[1:], capture_output=True, text=True, timeout=120)
            res[key] = json.loads(o.stdout.strip().splitlines()[-1]) if o.returncode == 0 else {"error": (o.stderr or o.stdout)[-200:]}
        except Exception as e:
            res[key] = {"error": str(e)[:200]}
    out = {"what": "ouster::sdk::core API on host containers, one 128x2048 dual-return frame per call (one PCIe round trip per call); "
                   "FrameStream: host packets in, XYZ of both returns out, batches of 32 frames, 3 in flight"}
    ha = res.get("host_api", {})
    if "ms_per_frame" in ha:
        m = ha["ms_per_frame"]
        out["frame_batcher_ms"] = m["FrameBatcher_128_packets"]
        out["frame_batcher_release_call_ms"] = m.get("FrameBatcher_release_call")
        out["destagger_ms"] = m["destagger_u32"]
        out["destagger_u8_ms"] = m.get("destagger_u8")
        out["xyzlut_ms"] = m["XYZLut_f64"]
        out["frame_total_ms"] = m.get("frame_total")
        out["frame_total_is"] = ha.get("frame_total_is")
        out["frame_matches_source"] = ha.get("frame_matches_source")
        out["allocations_in_timed_frames"] = ha.get("allocations_in_timed_frames")
        out["how"] = ("Field / img_t / PointCloudXYZ are pool (page-locked) memory: the decode, cartesian and inverse-destagger kernels "
                      "read and write them in place (one launch, no staging copy, no allocation); destagger() of a plane the "
                      "FrameBatcher has just released is one copy out of the HBM mirror its release launch left behind "
                      "and, once the batcher has seen which LUT the caller projects with, so is XYZLut()(frame) (DESIGN 4.1; link measurements: profiles/r06_dropin/copybench.json)")
    else:
        out["host_api_error"] = ha.get("error")
    fs = res.get("frame_stream", {})
    if "Mpoints_per_s" in fs:
        out["frame_stream_Gpoints_s"] = round(fs["Mpoints_per_s"] / 1e3, 3)
        out["frame_stream_H2D_GBps"] = fs.get("H2D_GBps")
        out["frame_stream_D2H_GBps"] = fs.get("D2H_GBps")
    else:
        out["frame_stream_error"] = fs.get("error")
    try:   # the same calls with the HBM mirror switched off: every destagger / XYZLut() is a kernel on the host containers in place
        o = subprocess.run([os.path.join(bdir, "bench_host_api"), "40"], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, OUSTER_HIP_MIRROR="0"))
        mo = json.loads(o.stdout.strip().splitlines()[-1])["ms_per_frame"] if o.returncode == 0 else None
        if mo:
            out["without_mirror"] = {"frame_batcher_ms": mo["FrameBatcher_128_packets"], "destagger_ms": mo["destagger_u32"],
                                     "destagger_u8_ms": mo.get("destagger_u8"), "xyzlut_ms": mo["XYZLut_f64"],
                                     "frame_total_ms": mo.get("frame_total"),
                                     "what": "OUSTER_HIP_MIRROR=0: nothing is served from what the release launch left in HBM"}
    except Exception as e:   # noqa: BLE001
        out["without_mirror"] = {"error": str(e)[:200]}
    try:   # the same three calls through the pybind11 module (numpy in, numpy out; results are numpy arrays over pool memory)
        o = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_python_api.py"), "30"], capture_output=True, text=True, timeout=180)
        pj = json.loads(o.stdout.strip().splitlines()[-1]) if o.returncode == 0 else {"error": (o.stderr or o.stdout)[-200:]}
    except Exception as e:   # noqa: BLE001
        pj = {"error": str(e)[:200]}
    out["python_core_ms"] = pj.get("ms_per_frame", pj)
    fc = res.get("frame_stream_compact", {})
    if "frames_per_s" in fc:   # round 6: the range-gated compacting route as what comes back (StreamOptions::dewarp_*)
        out["frame_stream_compact"] = {"frames_per_s": fc["frames_per_s"], "Gpixels_per_s": round(fc["Mpixels_per_s"] / 1e3, 3),
                                       "kept_points_per_frame": fc.get("kept_points_per_frame"), "H2D_GBps": fc.get("H2D_GBps"),
                                       "D2H_GBps": fc.get("D2H_GBps"), "frames_per_batch": 16, "in_flight": 4,
                                       "what": "host packets in, the gated (0.5 - 400 m) point list of the first return out: 12 B per kept "
                                               "point instead of 24 B per pixel; bound by the one host thread that stages the packets "
                                               "(its memcpy rate is H2D_GBps), no longer by the link from the device",
                                       "dense_xyz_frames_per_s": fs.get("frames_per_s")}
    elif fc:
        out["frame_stream_compact_error"] = fc.get("error")
    return out


def _report_row(fn, *a, **kw):
    """The rows beside the headline (loss paths, other workloads, latency, standalone kernels, drop-in API, CPU baseline) are
    evidence, never `value`: one of them failing must not cost the driver the line itself."""
    try:
        return fn(*a, **kw)
    except Exception as e:   # noqa: BLE001 -- reported in the line
        import traceback
        return {"error": f"{type(e).__name__}: {e}"[:300], "where": traceback.format_exc().strip().splitlines()[-3:]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=256, help="frames per step per GPU")
    ap.add_argument("--pool", type=int, default=16, help="distinct synthetic frames")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--time-every", type=int, default=4,
                    help="HIP events around every N-th decode kernel of the timed region (1: every launch)")
    ap.add_argument("--rotate-inputs", type=int, default=2,
                    help="decode this many copies of the packet batch in turn (>= 2: cold input every step)")
    ap.add_argument("--placement", default="refine", choices=["first", "refine", "draws"],
                    help="refine (default): the frugal placement search that DeviceFrameBatch runs by itself when it is constructed "
                         "(BatchOptions::auto_placement = true, the library's default since round 5): two more copies of the "
                         "output set back to back (6.6 GB transient, 0.1 s), each buffer group kept at the fastest of its three "
                         "locations; the first allocation's time is printed beside the kept one.  first: buffers as allocated "
                         "(auto_placement = false); the search is then run AFTER the timed region and reported beside the "
                         "headline.  draws: diagnostic, whole output sets drawn across the device memory")
    ap.add_argument("--placement-draws", type=int, default=3, help="--placement refine: locations tried per buffer group")
    ap.add_argument("--placement-ballast-gb", type=float, default=0.0,
                    help="--placement refine: device memory held between two locations (0: back-to-back draws; 8 with 4 draws "
                         "is round 3's 35 GB form)")
    ap.add_argument("--placement-scan-tries", type=int, default=10,
                    help="--placement refine: whole output sets (one allocation per array, back to back) drawn AFTER the group-wise search, "
                         "which stays a candidate (HotPath.pick_placement(slab=False) = DeviceFrameBatch::tune_placement).  About one process in "
                         "three gets nothing but slow placements for its first allocations and the three back-to-back draws of the frugal "
                         "search do not leave them (DESIGN.md 3.2).  0: the frugal search alone")
    ap.add_argument("--placement-scan-stride-gb", type=float, default=4.0, help="device memory held (allocated, never touched) between two draws of the scan")
    ap.add_argument("--placement-stride-gb", type=float, default=4.0,
                    help="--placement draws: ballast held between two draws (they scan the device memory)")
    ap.add_argument("--placement-tries", type=int, default=24,
                    help="--placement draws: candidate allocations of the output set (1 = the same as --placement first)")
    ap.add_argument("--no-loss-paths", action="store_true", help="skip the loss-path timings after the timed region")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the other BASELINE configs and the small-batch latency block after the timed region")
    ap.add_argument("--workload", default="dual", choices=sorted(WORKLOADS),
                    help="'dual' is the metric (configs[2]); the others are extra report rows")
    ap.add_argument("--outputs", default="full", choices=["full", "xyz", "planes", "planes+dst"],
                    help="ablation of the output set (the metric is 'full')")
    ap.add_argument("--pcie", action="store_true",
                    help="also time host->GPU packets + decode + GPU->host XYZ (reported separately)")
    ap.add_argument("--exchange", action="store_true",
                    help="also time RCCL scatter of packets / gather of XYZ (reported separately)")
    args = ap.parse_args()

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the hot path)")
    action, plan = plan_launch(args.gpus, os.environ, sys.argv[1:], torch.cuda.device_count())
    if action == "spawn":      # plain `python bench.py --gpus N`: start the N ranks, pass their one JSON line through
        import subprocess
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(plan, env=env))
    import torch.distributed as dist
    from ouster_sdk_amd.device import HotPath

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = plan
    # test knobs: BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and BENCH_BACKEND=gloo swaps RCCL
    # for gloo, so the N>1 launch path can be smoke-tested on a 1-GPU box
    one_device = os.environ.get("BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    torch.cuda.set_device(0 if one_device else local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    red_device = "cuda" if backend == "nccl" else "cpu"
    # how many ranks the collective library really joined (counted BY it: an all-reduce of ones), and which library
    rccl_ranks, coll = None, None
    if world > 1:
        ones = torch.ones(1, dtype=torch.int32, device=red_device)
        dist.all_reduce(ones)
        if backend == "nccl":
            rccl_ranks = int(ones.item())
            coll = "RCCL %s via torch.distributed 'nccl', one rank per GPU" % ".".join(str(x) for x in torch.cuda.nccl.version())
        else:
            coll = f"{backend} (test transport, {int(ones.item())} ranks): RCCL was not used"

    # The host-API rows (child processes: bench_host_api, bench_stream) are taken FIRST, while this process owns nothing on the GPU:
    # measured behind the placement search -- tens of device-to-device clones and tens of GB allocated and freed by this process --
    # the child's copy stream from the device ran at 27.8 GB/s instead of 50.7 (round 6; the headline and every resident row are
    # unaffected either way)
    drop_in_early = None
    if rank == 0 and world == 1 and args.workload == "dual" and args.outputs == "full" and not args.no_extras:
        drop_in_early = _report_row(time_drop_in)

    profile, bits, chan, dst_names, xyz_names, _, _, wl_label = WORKLOADS[args.workload]
    alt, az, shifts, b2l, l2s = synth_calibration()
    hp = HotPath(profile, H, W, CPP)
    hp.set_pixel_shift_by_row(shifts)
    lut_args = []
    if args.workload == "fused4":  # four sensors, four rigid extrinsics folded into their LUTs
        for k in range(4):
            a = 0.5 * np.pi * k
            ext = np.array([[np.cos(a), -np.sin(a), 0, 1000.0 * k], [np.sin(a), np.cos(a), 0, -500.0 * k],
                            [0, 0, 1, 250.0], [0, 0, 0, 1]])   # translation already in mm
            lut_args.append((b2l, ext @ l2s, az, alt))
    else:
        lut_args.append((b2l, l2s, az, alt))
    for la in lut_args:
        hp.add_lut(*la)

    pool = synth_packets(args.pool, seed=0xDEADBEEF + 1000 * rank, bits=bits, chan=chan)
    F = args.frames
    if args.workload == "batch512":      # fixed total work, split over the ranks
        F = 512 // world
    d_pool = torch.from_numpy(pool).cuda()
    packets = d_pool.repeat((F + args.pool - 1) // args.pool, 1, 1)[:F].contiguous()
    def make_outputs():
        if args.outputs == "full":
            return hp.alloc_outputs(F, destagger=dst_names, xyz=xyz_names)
        if args.outputs == "xyz":
            return hp.alloc_outputs(F, planes=[], xyz=xyz_names, headers=False)
        if args.outputs == "planes":
            return hp.alloc_outputs(F)
        return hp.alloc_outputs(F, destagger=dst_names)
    out = make_outputs()

    def barrier():
        if world > 1:
            dist.barrier()

    # setup, like building the LUT: let the library's per-workload kernel-variant tuner finish
    # (it times up to five kernel variants four times each on the first twenty calls), so the W
    # warm-up steps and the K timed steps all run the variant it settled on
    for _ in range(24):
        hp.decode(packets, out)
    torch.cuda.synchronize()
    # setup, like sizing a memory pool: the physical placement of a buffer is drawn when it is allocated and
    # the decode's write rate differs by 10 - 20 % between draws (tools/ab/alloc_lottery.py).
    # A pipeline allocates its buffers once, so it can afford to draw a few and keep the best -- that is what
    # HotPath.pick_placement does.  Every draw's time is reported ("placement"); --placement-tries 1 turns it off.
    placement = None
    if args.placement == "draws" and args.placement_tries > 1:
        del out
        torch.cuda.empty_cache()
        packets, out, placement = hp.pick_placement(packets, make_outputs, tries=args.placement_tries,
                                                    stride_gb=args.placement_stride_gb)
        placement["mode"] = "draws"
    elif args.placement == "refine":
        # the search holds (draws - 1) more copies of the output set (+ ballast): not where the device memory left to this rank
        # is short (N ranks of a test on one device; a shared GPU) -- then the first allocation stands and the line says so
        out_bytes = sum(v.numel() * v.element_size() for v in out.values())
        need = (args.placement_draws - 1) * (out_bytes + int(args.placement_ballast_gb * (1 << 30)))
        free_b, total_b = torch.cuda.mem_get_info()
        share = free_b // (world if one_device else 1)
        if share < need + need // 4 + (2 << 30):
            placement = {"mode": "first", "skipped": f"placement search needs {need >> 20} MB transient, {share >> 20} MB of device "
                                                     f"memory free for this rank ({free_b >> 20} MB free in all): first allocation kept"}
        else:
            t_setup = time.perf_counter()
            out, placement = hp.refine_placement(packets, out, draws=args.placement_draws, ballast_gb=args.placement_ballast_gb)
            placement["mode"] = "refine"
            # then whole output sets, one allocation per array, back to back (HotPath.pick_placement(slab=False) = what
            # DeviceFrameBatch::tune_placement does): a process whose first allocations all landed badly -- about one in three --
            # usually has a good set among its next ten; what the group-wise search kept is a candidate
            per_draw = out_bytes + int(args.placement_scan_stride_gb * (1 << 30))
            tries = min(args.placement_scan_tries, int(share * 0.6) // per_draw - 1) if world == 1 or not one_device else 0
            if tries >= 2:
                try:
                    packets, out2, scan = hp.pick_placement(packets, make_outputs, tries=tries, launches=8,
                                                            stride_gb=args.placement_scan_stride_gb, incumbent=out, slab=False)
                    placement["scan"] = scan
                    out = out2
                    # ... and the group-wise search once more: its copies now land in the memory the scan has just given back,
                    # which is where a process whose fresh allocations were all slow found its fast groups
                    # (profiles/r06_latency/placement_notes.txt)
                    out, rep2 = hp.refine_placement(packets, out, draws=args.placement_draws, ballast_gb=args.placement_ballast_gb)
                    placement["after_scan"] = {k: rep2[k] for k in ("first_allocation_ms", "kept_ms", "groups")}
                    placement["kept_ms"] = rep2["kept_ms"]
                except Exception as e:   # what the group-wise search kept stands
                    placement["scan"] = {"error": str(e)[:200]}
                    torch.cuda.empty_cache()
            placement["setup_s"] = round(time.perf_counter() - t_setup, 3)
    for _ in range(args.warmup):
        hp.decode(packets, out)
    torch.cuda.synchronize()
    barrier()
    # --rotate-inputs R > 1 (a diagnostic, not the metric): R copies of the packet batch are decoded in turn, so that
    # no step finds its input in the 256 MB Infinity Cache left there by the step before
    inputs = [packets] + [packets.clone() for _ in range(max(0, args.rotate_inputs - 1))]
    hp.ctx.timing(args.time_every)   # HIP events around every N-th decode kernel of the timed region (each pair costs the stream 2 - 3 us)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        hp.decode(inputs[i % len(inputs)], out)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    kern_ms, n_launch = hp.ctx.timing_read()
    tc, tr = hp.ctx.last_decode_tile()
    spec_name = "SpecSingle" if args.workload in ("single", "batch512") else "SpecDualLB"
    kern = hp.ctx.last_decode_kernel() or ("k_decode_wide" if tr < H else "k_decode")
    kernel_name = f"{kern}<{spec_name},{tc},sep-f32> ({tc}x{tr} tiles)"
    tuner = {"source": hp.ctx.last_decode_tuner(), "kernel": kern, "tile": [tc, tr],
             "cache_file": os.environ.get("OUSTER_HIP_TUNING_CACHE"),
             "note": "measured: this process timed the candidates during setup (24 calls before the warm-up); cache: read from "
                     "OUSTER_HIP_TUNING_CACHE / BatchOptions::tuning_cache, nothing timed (tests/test_gpu_tuning_cache.py)"}
    hp.ctx.timing(False)


    # context for the roofline: what a plain device-to-device copy reaches on THIS box right now
    # (boxes of the pool differ by up to ~35 % in sustained HBM bandwidth)
    cp_src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    cp_dst = torch.empty_like(cp_src)
    cp_dst.copy_(cp_src)
    torch.cuda.synchronize()
    c0 = time.perf_counter()
    for _ in range(5):
        cp_dst.copy_(cp_src)
    torch.cuda.synchronize()
    box_copy_gbps = 5 * 2 * (1 << 30) / (time.perf_counter() - c0) / 1e9
    del cp_src, cp_dst

    pcie = None
    if args.pcie:
        # PCIe-inclusive rate (never the headline `value`): pinned host packets in, XYZ of both
        # returns back out, same kernels in between, no overlap between the three stages
        nf = min(F, 64)
        h_in = torch.empty((nf,) + tuple(packets.shape[1:]), dtype=torch.uint8).pin_memory()
        h_in.copy_(packets[:nf].cpu())
        d_in = torch.empty_like(packets[:nf])
        o2 = hp.alloc_outputs(nf, planes=[], xyz=xyz_names, headers=False)
        h_out = {k: torch.empty(v.shape, dtype=v.dtype).pin_memory() for k, v in o2.items()}

        def one():
            d_in.copy_(h_in, non_blocking=True)
            hp.decode(d_in, o2)
            for k in o2:
                h_out[k].copy_(o2[k], non_blocking=True)
        one(); torch.cuda.synchronize()
        p0 = time.perf_counter()
        for _ in range(5):
            one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - p0) / 5
        moved = h_in.numel() + sum(v.numel() * v.element_size() for v in h_out.values())
        pcie = {"Mpoints_per_s": round(nf * H * W * len(xyz_names) / dt / 1e6, 1),
                "frames": nf, "GBps_over_pcie": round(moved / dt / 1e9, 1),
                "what": "pinned H2D packets + decode + D2H XYZ f32, serialized"}

    elapsed = t1 - t0
    per_rank = None
    if world > 1:
        # every rank's own view next to the max the line is computed from: a slow rank, or one whose tuner settled on another
        # kernel, must be visible in a SCALE record (rank 0 would otherwise speak for all of them)
        # (plain tensors through the same all-gather path as the reduction below: no pickling, nothing RCCL has not done before)
        KERNELS = ["k_decode", "k_decode_wide", "k_decode_stream", "k_decode_stream2", "other"]
        pl_mode = (placement or {}).get("mode", "first")
        pl_code = {"first": 0, "refine": 1, "draws": 2}.get(pl_mode, 0) + (10 if (placement or {}).get("skipped") else 0)
        mine_t = torch.tensor([(t1 - t0) / args.steps * 1e3, kern_ms, float(KERNELS.index(kern) if kern in KERNELS else 4), float(tc), float(tr),
                               float(pl_code), float(torch.cuda.current_device())], dtype=torch.float64, device=red_device)
        all_t = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(all_t, mine_t)
        rows = []
        for r_i, v in enumerate(all_t):
            v = v.cpu().tolist()
            pc = int(v[5])
            rows.append({"rank": r_i, "ms_per_step": round(v[0], 4), "kernel_ms": round(v[1], 4),
                         "kernel": f"{KERNELS[int(v[2])]} {int(v[3])}x{int(v[4])}", "device": int(v[6]),
                         "placement": ["first", "refine", "draws"][pc % 10] + (":skipped" if pc >= 10 else "")})
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        kernels = {}
        for r_ in rows:
            kernels[r_["kernel"]] = kernels.get(r_["kernel"], 0) + 1
        per_rank = {"ms_min": min(r_["ms_per_step"] for r_ in rows), "ms_max": max(r_["ms_per_step"] for r_ in rows),
                    "kernel_ms_min": min(r_["kernel_ms"] for r_ in rows), "kernel_ms_max": max(r_["kernel_ms"] for r_ in rows),
                    "kernel": kernels, "slowest_rank": max(rows, key=lambda r_: r_["ms_per_step"])["rank"],
                    "placement": sorted({r_["placement"] for r_ in rows}),
                    "note": "ms_* are each rank's own timed region (its barrier wait included); `ms_per_step` of the line is the max"}

    exchange = None
    if args.exchange:
        # the only real exchange step of the path: the batch of all ranks' frames lives on
        # rank 0, every rank receives ITS shard (grouped point-to-point sends, one peer per xGMI link),
        # decodes it, and the XYZ clouds are gathered back.  Reported per stage, never part of `value`.
        # `xyz_checksum` (sum of the gathered clouds' bit patterns) does not depend on the number of ranks: with one rank the
        # same batch is decoded in place (tests/test_gpu_multirank.py compares the two).
        from ouster_sdk_amd import parallel
        xdev = "cuda" if backend == "nccl" else "cpu"
        total = F * world
        batch_all = packets.repeat(world, 1, 1).to(xdev) if rank == 0 else None
        ts = [0.0, 0.0, 0.0]
        checksum = 0
        for rep in range(3):                       # rep 0 warms the RCCL channels
            torch.cuda.synchronize(); barrier()
            e0 = time.perf_counter()
            if world > 1:
                mine = parallel.scatter_frames(batch_all, total, tuple(packets.shape[1:]), torch.uint8, xdev).cuda()
            else:
                mine = batch_all.cuda()
            torch.cuda.synchronize(); barrier()
            e1 = time.perf_counter()
            hp.decode(mine, out)
            torch.cuda.synchronize(); barrier()
            e2 = time.perf_counter()
            gathered = []
            for n in xyz_names:
                x = out["xyz:" + n]
                gathered.append(parallel.gather_frames(x if backend == "nccl" else x.cpu(), total) if world > 1 else x)
            torch.cuda.synchronize(); barrier()
            e3 = time.perf_counter()
            if rep:
                ts = [ts[0] + e1 - e0, ts[1] + e2 - e1, ts[2] + e3 - e2]
            if rep == 2 and rank == 0:
                checksum = sum(int(g.contiguous().view(torch.int32).to(torch.int64).sum().item()) for g in gathered) & ((1 << 63) - 1)
        del batch_all
        pk_bytes = total * packets[0].numel() * (world - 1) / world
        xyz_bytes = len(xyz_names) * total * H * W * 12 * (world - 1) / world
        exchange = {"frames_total": total, "scatter_packets_ms": round(ts[0] / 2 * 1e3, 3),
                    "decode_ms": round(ts[1] / 2 * 1e3, 3), "gather_xyz_ms": round(ts[2] / 2 * 1e3, 3),
                    "scatter_GBps": round(pk_bytes / (ts[0] / 2) / 1e9, 1),
                    "gather_GBps": round(xyz_bytes / (ts[2] / 2) / 1e9, 1),
                    "Mpoints_per_s_with_exchange": round(total * H * W * len(xyz_names) / (sum(ts) / 2) / 1e6, 1),
                    "xyz_checksum": checksum,
                    "transport": ("RCCL p2p (xGMI)" if backend == "nccl" else backend + " (test transport)") if world > 1 else "none (one rank)"}

    n_ret = len(xyz_names)
    points_per_step = F * H * W * n_ret * world
    value = points_per_step * args.steps / elapsed / 1e6
    bytes_per_launch = algorithmic_bytes_per_frame(args.workload) * F
    if args.outputs != "full":  # ablation runs: count only what is actually moved
        per = (W // CPP) * (32 + CPP * (12 + H * 8) + 32)
        per += {"xyz": 2 * H * W * 12, "planes": H * W * 15, "planes+dst": H * W * 25}[args.outputs]
        bytes_per_launch = per * F
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    step_gbps = bytes_per_launch * args.steps / elapsed / 1e9   # whole step: both passes + launch gaps

    # self-check of what the timed steps left in HBM (rank 0, outside the timed region)
    validated, max_dxyz, checked = None, None, None
    if rank == 0 and args.outputs == "full":     # (--no-cpu skips the CPU baseline leg only)
        try:
            validated, max_dxyz, checked = validate_against_oracle(
                hp, profile, packets, out, shifts, lut_args, len(lut_args), sorted({0, 7 % F, F // 2, F - 1}))
        except Exception as e:   # noqa: BLE001 -- the checker itself broke (not a mismatch): say so in the line, keep the line
            validated, max_dxyz, checked = None, None, [f"validation did not run: {type(e).__name__}: {e}"[:200]]
    # what the opt-in placement search (BatchOptions::auto_placement / HotPath.refine_placement) would have
    # given on this box: run AFTER the timed region and the self-check, reported beside the headline, never as `value`
    if rank == 0 and world == 1 and args.placement == "first" and args.outputs == "full" and not args.no_extras:
        t_setup = time.perf_counter()
        try:
            out, rep = hp.refine_placement(packets, out, draws=args.placement_draws, ballast_gb=args.placement_ballast_gb)
            placement = {"mode": "first", "search_after_timed_region": rep, "search_s": round(time.perf_counter() - t_setup, 3)}
        except Exception as e:   # noqa: BLE001
            placement = {"mode": "first", "search_error": str(e)[:200]}
    # the paths the metric never touches, on the same output buffers, outside the timed region and
    # after the self-check (they overwrite the outputs)
    loss_paths = None
    if rank == 0 and not args.no_loss_paths and args.outputs == "full":
        loss_paths = _report_row(time_loss_paths, hp, packets, out, F, algorithmic_bytes_per_frame(args.workload))
    # every other BASELINE config and the small-batch latency view, in the driver-visible line
    other_workloads, latency, latency_python, standalone, drop_in = None, None, None, None, None
    if rank == 0 and world == 1 and args.workload == "dual" and args.outputs == "full" and not args.no_extras:
        other_workloads = _report_row(time_other_workloads, placement="first" if args.placement == "first" else "refine",
                                      scan_tries=args.placement_scan_tries, scan_stride_gb=args.placement_scan_stride_gb)   # same default
        latency_python = _report_row(time_small_batches)
        latency = _report_row(time_latency_cpp, latency_python)
        if "error" in latency:      # the C++ helper is missing: the Python-timed rows are the only ones
            latency = latency_python
        torch.cuda.empty_cache()
        standalone = _report_row(time_standalone)
        drop_in = drop_in_early
    # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot wrap a run from inside):
    # only quoted when the committed profile is of exactly this workload and output set.
    traffic, traffic_src = None, None
    if args.outputs == "full":
        import glob
        cands = []
        for pj in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                "profiles", "*", "pmc_traffic.json")), reverse=True):
            try:
                cands.append((json.load(open(pj)).get("kernel_sources_sha256") != kernel_sources_sha256(), pj))
            except Exception:
                pass
        # counters recorded on THIS tree's kernel sources first; among those the round's own directory ("r03") before its
        # side runs ("r03_slowbox", ...); then the most recent directory name
        for _, pj in sorted(cands, key=lambda c: (c[0], len(os.path.basename(os.path.dirname(c[1]))) if not c[0] else 0)):
            try:
                t = json.load(open(pj))
                if t.get("workload") == args.workload and t.get("frames_per_launch"):
                    # the two persistent kernels (k_decode_stream: every wave fetches; k_decode_stream2: loader waves) share
                    # one key in the committed counters; the entry names the one that was profiled
                    fam = "k_decode_stream" if kern.startswith("k_decode_stream") else kern
                    v = t.get("variants", {}).get(f"{fam}:{tc}") or \
                        (t.get("variants_by_tile_columns", {}).get(str(tc)) if fam != "k_decode_stream" else None)
                    if v and fam == "k_decode_stream" and (kern + "<") not in v.get("kernel", kern + "<"):
                        continue           # counters of the sibling persistent kernel: not this run's
                    if not v:
                        continue           # no committed counters for the variant that ran here
                    traffic = int(round(v["total_bytes"] * F / t["frames_per_launch"]))
                    stale = t.get("kernel_sources_sha256") != kernel_sources_sha256()
                    traffic_src = (os.path.relpath(pj, os.path.dirname(os.path.abspath(__file__))) +
                                   f": {v['total_bytes']} B per {t['frames_per_launch']}-frame launch of " +
                                   f"{v.get('kernel', t.get('kernel'))}; " + t.get("method", "") +
                                   ("; RECORDED ON OTHER KERNEL SOURCES than this tree's (stale)" if stale else
                                    "; recorded on this tree's kernel sources"))
                    break
            except Exception:
                pass

    if rank == 0:
        line = {
            "metric": "Mpoints/sec projected (decode+destagger+cartesian), 128x2048 dual-return",
            "value": round(value, 1), "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if args.workload == "batch512" else "weak",
            "vs_baseline": None, "dtype": "u32+f64",
            "data": "synthetic",
            "config": {"workload": wl_label,
                       "frames_per_step_per_gpu": F, "points_per_frame": H * W * n_ret,
                       "outputs": (f"{len(hp.fields)} planes + {len(dst_names)} destaggered planes + "
                                   f"{len(xyz_names)}x XYZ f32 + column headers")
                       if args.outputs == "full" else "ABLATION:" + args.outputs,
                       "sharding": f"frames x{world}, no data-path collective",
                       "input_batches_rotated": args.rotate_inputs,
                       "buffer_placement": ("first allocation: the buffers as the allocator returned them "
                                            "(BatchOptions::auto_placement = false)"
                                            if (not placement or placement.get("mode") == "first") else
                                            ((("each buffer group at the fastest of %d back-to-back locations, then the fastest of that and %d further "
                                               "whole output sets and of up to 10 locations of the packet buffer (HotPath.pick_placement(slab=False) = "
                                               "DeviceFrameBatch::tune_placement, a setup-time option of the library: about one process in three gets only "
                                               "slow placements for its first allocations, DESIGN.md 3.2); with the library's default alone: "
                                               % (placement["draws_per_group"], len(placement["scan"].get("output_sets_ms", []))))
                                              if placement.get("scan") and "error" not in placement["scan"] else "") +
                                             "each buffer group at the fastest of %d locations %g GB apart (HotPath.refine_placement = what "
                                             "DeviceFrameBatch does at construction with BatchOptions::auto_placement = true, the library's default); "
                                             "roofline.first_allocation_* is the same run on the buffers as the allocator first returned them"
                                             % (placement["draws_per_group"], args.placement_ballast_gb)) if placement["mode"] == "refine" else
                                            "DIAGNOSTIC: best of %d allocations of the output set (%.0f GB of ballast between "
                                            "two draws) and of up to 10 of the packet buffer (HotPath.pick_placement)"
                                            % (args.placement_tries, args.placement_stride_gb))},
            "placement": placement,
            "roofline": {"bound": "hbm",
                         "kernel": kernel_name,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "achieved_step": round(step_gbps, 1),
                         "frac_step": round(step_gbps / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "kernel_ms_avg": round(kern_ms, 4), "launches_timed": n_launch,
                         "box_d2d_copy_GBps": round(box_copy_gbps, 1)},
            "validated": validated, "max_abs_dxyz_m": max_dxyz, "validated_frames": checked,
            "loss_paths": loss_paths,
            "other_workloads": other_workloads,
            "latency": latency,
            "latency_python": latency_python,
            "standalone": standalone,
            "drop_in": drop_in,
            "cpu_baseline": None,
        }
        if placement and "first_allocation_ms" in placement:   # --placement refine: the first allocation's fraction next to the kept draw's
            line["roofline"]["first_allocation_ms_per_call"] = placement["first_allocation_ms"]
            line["roofline"]["first_allocation_frac_step"] = round(
                bytes_per_launch / (placement["first_allocation_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
        if placement and isinstance(placement.get("scan"), dict) and placement["scan"].get("incumbent_ms"):
            # ... and what the library's default search alone had reached before the whole sets were drawn (same clock: 8 back-to-back
            # decodes into the buffers it kept) -- three numbers for one run: first allocation, library default, this setup
            line["roofline"]["library_default_search_ms_per_call"] = placement["scan"]["incumbent_ms"]
            line["roofline"]["library_default_search_frac_step"] = round(
                bytes_per_launch / (placement["scan"]["incumbent_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
        if placement and "search_after_timed_region" in placement:   # default: what the opt-in search would have given
            rep = placement["search_after_timed_region"]
            line["roofline"]["searched_placement_ms_per_call"] = rep["kept_ms"]
            line["roofline"]["searched_placement_frac_step"] = round(bytes_per_launch / (rep["kept_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
        line["tuner"] = tuner
        line["rccl_ranks"] = rccl_ranks
        line["collective_backend"] = coll
        if per_rank:
            line["per_rank"] = per_rank
        if exchange:
            line["exchange"] = exchange
        if pcie:
            line["pcie_inclusive"] = pcie
        if world == 1 and not args.no_cpu and args.workload == "dual":
            line["cpu_baseline"] = _report_row(cpu_baseline, pool, shifts)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
