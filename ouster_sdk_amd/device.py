"""torch-backed device buffers around the C ABI (plumbing: memory, streams, distribution).

`HotPath` bundles a context, a format and optional LUTs for one sensor configuration and
exposes the three operations of the path on torch CUDA(=HIP) tensors:

    decode(packets)   -> dict of planes / destaggered planes / xyz / column headers
    destagger(img)    -> destaggered image(s)
    cartesian(range)  -> xyz

All arithmetic happens in libouster_hip.so; torch only owns the HBM allocations and the
stream the kernels are ordered on.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _capi as capi

_TORCH_OF_ELEM = {1: torch.uint8, 2: torch.uint16, 4: torch.uint32, 8: torch.uint64}


def _u8(n):
    return torch.empty(n, dtype=torch.uint8, device="cuda")


class HotPath:
    def __init__(self, profile: str, h: int, w: int, cpp: int = 16, header_type: int = 0,
                 fields: Optional[Sequence[Tuple[str, int]]] = None, with_window: bool = True,
                 device: Optional[int] = None, use_torch_stream: bool = True, lib=None):
        if not torch.cuda.is_available():
            raise capi.OusterHipError("no MI355X visible: the hot path has no CPU fallback")
        self.device = torch.cuda.current_device() if device is None else device
        torch.cuda.set_device(self.device)
        if os.environ.get("OUSTER_HIP_OWN_STREAM"):  # experiment knob
            use_torch_stream = False
        stream = torch.cuda.current_stream().cuda_stream if use_torch_stream else None
        self.ctx = capi.Context(self.device, stream, lib=lib)   # lib: a privately loaded A/B build (capi.load_hip(path))
        self.profile, self.h, self.w, self.cpp = profile, h, w, cpp
        self.fields: List[Tuple[str, int]] = list(fields) if fields is not None else \
            capi.default_planes(profile, with_window)
        self.desc = capi.format_desc(profile, h, cpp, w, self.fields, header_type)
        self.fmt = self.ctx.make_format(self.desc)
        self.packet_size = self.desc.lidar_packet_size
        self.luts: List[capi.Lut] = []
        self.shifts: Optional[np.ndarray] = None

    # -- configuration -----------------------------------------------------------------
    def set_pixel_shift_by_row(self, shifts):
        self.shifts = np.ascontiguousarray(shifts, dtype=np.int32)

    def add_lut(self, beam_to_lidar, transform, az_deg, alt_deg, range_unit=0.001) -> capi.Lut:
        lut = capi.Lut.from_calib(self.ctx, self.w, self.h, range_unit, beam_to_lidar, transform,
                                  az_deg, alt_deg)
        self.luts.append(lut)
        return lut

    def add_lut_arrays(self, direction, offset) -> capi.Lut:
        lut = capi.Lut.from_arrays(self.ctx, direction, offset, self.h, self.w)
        self.luts.append(lut)
        return lut

    def field_index(self, name: str) -> int:
        for i, (n, _) in enumerate(self.fields):
            if n == name:
                return i
        return -1

    # -- output allocation ---------------------------------------------------------------
    def alloc_outputs(self, n_frames: int, planes: Optional[Sequence[str]] = None,
                      destagger: Sequence[str] = (), xyz: Sequence[str] = (),
                      xyz_dtype=torch.float32, headers: bool = True) -> Dict[str, torch.Tensor]:
        out: Dict[str, torch.Tensor] = {}
        names = [n for n, _ in self.fields] if planes is None else list(planes)
        for n, es in self.fields:
            if n in names:
                shape = (n_frames, self.h, self.w) if es != 6 else (n_frames, self.h, self.w, 3)
                dt = _TORCH_OF_ELEM[es] if es != 6 else torch.uint16
                out[n] = torch.empty(shape, dtype=dt, device="cuda")
            if n in destagger:
                shape = (n_frames, self.h, self.w) if es != 6 else (n_frames, self.h, self.w, 3)
                dt = _TORCH_OF_ELEM[es] if es != 6 else torch.uint16
                out["destaggered:" + n] = torch.empty(shape, dtype=dt, device="cuda")
        for n in xyz:
            out["xyz:" + n] = torch.empty((n_frames, self.h * self.w, 3), dtype=xyz_dtype,
                                          device="cuda")
        if headers:
            out["timestamp"] = torch.empty((n_frames, self.w), dtype=torch.uint64, device="cuda")
            out["measurement_id"] = torch.empty((n_frames, self.w), dtype=torch.uint16, device="cuda")
            out["status"] = torch.empty((n_frames, self.w), dtype=torch.uint32, device="cuda")
            out["frame_meta"] = torch.empty((n_frames, 24), dtype=torch.uint8, device="cuda")
        return out

    def pick_placement(self, packets: torch.Tensor, make_outputs, tries: int = 16, launches: int = 12,
                       stride_gb: float = 0.0, incumbent: Optional[Dict[str, torch.Tensor]] = None, slab: bool = True):
        """Output buffers that live for the life of a pipeline are worth choosing: on MI355X the achieved
        write rate of the decode differs by 10 - 20 % between allocations of the SAME size made by the SAME
        process (tools/ab/alloc_lottery.py: the physical placement of an allocation is drawn when it is made
        and never changes), in two modes -- most draws land near the slow one.  This helper draws `tries`
        candidate output sets with make_outputs() (each one slab, one draw), times the decode into each and
        keeps the fastest; then does the same for copies of the packet buffer.  Returns
        (packets, outputs, report).  Everything but the winners is freed.
        stride_gb > 0: hold that much ballast between two draws, so that `tries` draws scan the device memory
        instead of its first tries x (set size) -- which mode a draw gets follows where it lands
        (tools/ab/ballast.py), and the fast regions can be tens of GB apart.
        incumbent: an output set the caller already has; it is timed first and is a candidate like every draw (the
        result is then never slower than what the caller came with; report["incumbent_ms"]).
        slab=False: a draw is make_outputs() as it is (one allocation per array) -- round 6: one slab per set is one physical
        draw, but slabs themselves tend to be slower than separately allocated arrays (tools/ab/alloc_lottery.py slab, profiles/r06_latency/placement_notes.txt)."""
        def slab_set():
            tmpl = make_outputs()
            names = list(tmpl)
            sizes = [tmpl[n].numel() * tmpl[n].element_size() for n in names]
            meta = {n: (tmpl[n].dtype, tuple(tmpl[n].shape)) for n in names}
            del tmpl
            al = 2 << 20
            slab = torch.empty(sum((x + al - 1) // al * al for x in sizes) + al, dtype=torch.uint8, device="cuda")
            off = (-slab.data_ptr()) % al
            out = {}
            for n, nb in zip(names, sizes):
                out[n] = slab[off:off + nb].view(meta[n][0]).view(meta[n][1])
                off += (nb + al - 1) // al * al
            return out

        def clock(pk, out):
            for _ in range(2):
                self.decode(pk, out)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(launches):
                self.decode(pk, out)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / launches

        # the rejected draws stay allocated until the search is over: memory that has just been freed is what
        # the next allocation of the same size gets back, and it would be the same draw again
        best_out, best_ms, out_ms, held = None, None, [], []
        incumbent_ms = None
        if incumbent is not None:
            incumbent_ms = clock(packets, incumbent)
            best_out, best_ms = incumbent, incumbent_ms
            if stride_gb > 0:
                try:
                    held.append(torch.empty(int(stride_gb * (1 << 30)), dtype=torch.uint8, device="cuda"))
                except RuntimeError:
                    pass
        for _ in range(max(1, tries) if incumbent is None else max(0, tries)):
            try:
                cand = slab_set() if slab else make_outputs()
            except RuntimeError:        # out of device memory: choose among the draws made so far
                if best_out is None:
                    raise
                break
            held.append(cand)
            ms = clock(packets, cand)
            out_ms.append(round(ms, 4))
            if best_ms is None or ms < best_ms:
                best_out, best_ms = cand, ms
            if stride_gb > 0:
                try:
                    held.append(torch.empty(int(stride_gb * (1 << 30)), dtype=torch.uint8, device="cuda"))
                except RuntimeError:
                    break
        # the outputs are chosen: release the other draws and the ballast, then look for the packet buffer's
        # place the same way (it is small, so the ballast between its draws is larger)
        held = []
        torch.cuda.empty_cache()
        best_pk, pk_best_ms, pk_ms = packets, best_ms, [round(best_ms, 4)]
        cand = None
        n_pk = max(0, min(tries, 10) - 1)
        for _ in range(n_pk):
            try:
                cand = packets.clone()
            except RuntimeError:
                break
            held.append(cand)
            ms = clock(cand, best_out)
            pk_ms.append(round(ms, 4))
            if ms < pk_best_ms:
                best_pk, pk_best_ms = cand, ms
            if stride_gb > 0:
                try:
                    held.append(torch.empty(int(5 * stride_gb * (1 << 30)), dtype=torch.uint8, device="cuda"))
                except RuntimeError:
                    break
        del held, cand
        torch.cuda.empty_cache()
        # the kernel variant was chosen on the buffers the caller had before: let the tuner look again
        try:
            self.ctx.set_knob("retune", 1)
        except (capi.OusterHipError, AttributeError):   # an older A/B build of the library without that knob
            pass
        else:
            for _ in range(24):
                self.decode(best_pk, best_out)
            torch.cuda.synchronize()
        rep = {"tries": tries, "stride_gb": stride_gb, "output_sets_ms": out_ms, "packet_buffers_ms": pk_ms}
        if incumbent_ms is not None:
            rep["incumbent_ms"] = round(incumbent_ms, 4)
        return best_pk, best_out, rep

    def refine_placement(self, packets: torch.Tensor, out: Dict[str, torch.Tensor], draws: int = 4, launches: int = 10,
                         ballast_gb: float = 8.0):
        """Where the output buffers live, settled group by group (the cheap successor of pick_placement; what
        ouster::sdk::hip::DeviceFrameBatch does by itself when it is constructed).

        The allocation lottery is mostly an interaction between a few heavy output streams
        (tools/ab/hybrid_sets.py: exchanging the two XYZ buffers of a slow set for those of a fast one recovers 90 % of
        the difference, exchanging one buffer alone changes nothing), and fast and slow regions of the device memory
        are tens of GB wide (tools/ab/ballast.py).  So: `draws - 1` further copies of the output set are allocated,
        `ballast_gb` of device memory held between two of them (they land in different regions); then, group by group
        -- XYZ clouds, 32-bit planes, destaggered planes, narrow planes -- the decode is timed with that group's
        buffers taken from each copy in turn and the fastest location is kept (coordinate descent; the buffers the
        caller came with are a candidate, so the result is never slower).  Everything else is freed.
        Transient footprint (draws - 1) x (output set + ballast), ~35 GB for 256 dual-return frames with the defaults;
        setup well under a second of allocations plus groups x draws x launches decodes.  ballast_gb = 0, draws = 3 is
        the footprint-frugal form (back-to-back draws share a region: it finds a fast place about every other time).
        Returns (out, report); `out` is updated in place."""
        def clock(o):
            for _ in range(2):
                self.decode(packets, o)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(launches):
                self.decode(packets, o)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / launches

        def group_of(k, t):
            if k.startswith("xyz:"):
                return "xyz"
            if t.dim() < 3:
                return None                      # column headers / frame meta: a few KB
            if k.startswith("destaggered:"):
                return "destaggered"
            return "planes32" if t.element_size() >= 4 else "planes8_16"

        groups: Dict[str, List[str]] = {}
        for k, t in out.items():
            g = group_of(k, t)
            if g:
                groups.setdefault(g, []).append(k)
        order = [g for g in ("xyz", "planes32", "destaggered", "planes8_16") if g in groups]
        keys = [k for g in order for k in groups[g]]
        t_alloc = __import__("time").perf_counter()
        copies, ballast = [], []
        for _ in range(max(0, draws - 1)):
            try:
                if ballast_gb > 0:
                    ballast.append(torch.empty(int(ballast_gb * (1 << 30)), dtype=torch.uint8, device="cuda"))
                copies.append({k: torch.empty_like(out[k]) for k in keys})
            except RuntimeError:                 # out of device memory: choose among what has been drawn
                break
        torch.cuda.synchronize()
        alloc_s = __import__("time").perf_counter() - t_alloc
        first_ms = best_ms = clock(out)
        report = {"first_allocation_ms": round(first_ms, 4), "draws_per_group": 1 + len(copies),
                  "ballast_gb_between_draws": ballast_gb, "allocation_s": round(alloc_s, 3), "groups": {}}
        for g in order:
            times = []
            for cand in copies:
                trial = dict(out)
                trial.update({k: cand[k] for k in groups[g]})
                times.append(clock(trial))
            report["groups"][g] = [round(x, 4) for x in times]
            if times and min(times) < best_ms:
                best_ms = min(times)
                out.update({k: copies[int(np.argmin(times))][k] for k in groups[g]})
        del copies, ballast
        torch.cuda.empty_cache()
        report["kept_ms"] = round(best_ms, 4)
        try:
            self.ctx.set_knob("retune", 1)
        except (capi.OusterHipError, AttributeError):
            pass
        else:
            for _ in range(24):
                self.decode(packets, out)
            torch.cuda.synchronize()
        return out, report

    # -- the three operations --------------------------------------------------------------
    def range_gate(self, min_range: float, max_range: float) -> Tuple[int, int, bool]:
        """(min_r, max_r, empty): the raw gate ouster_hip_dewarp_frames derives from metres."""
        lo, hi, e = C.c_uint32(), C.c_uint32(), C.c_int()
        capi.check(self.ctx.L.ouster_hip_range_gate(float(min_range), float(max_range), C.byref(lo), C.byref(hi),
                                                    C.byref(e)))
        return lo.value, hi.value, bool(e.value)

    def decode(self, packets: torch.Tensor, out: Dict[str, torch.Tensor],
               packet_counts=None,
               host_timestamps: Optional[torch.Tensor] = None,
               gate: Optional[Tuple[float, float]] = None, gate_field: str = "RANGE",
               poses: Optional[torch.Tensor] = None):
        """packets: uint8 CUDA tensor [n_frames, slots, packet_stride].
        gate = (min_range, max_range) in metres: also produce out["gate_counts"] (u16 [n, 8, W]), the
        per-column kept counts a following dewarp_frames(..., gate_counts=...) with the same gate needs.
        poses: float64 CUDA tensor [n_frames, W, 4, 4] (per-column body_to_world): the xyz outputs become
        dewarp(cartesian(range), poses), computed while the points are in registers."""
        assert packets.is_cuda and packets.dtype == torch.uint8 and packets.is_contiguous()
        n_frames, slots, stride = packets.shape
        fo = capi.FrameOut()
        fo.xyz_field[0] = fo.xyz_field[1] = -1
        fo.xyz_dtype = capi.F32
        any_dst = False
        xyz_names = [k[4:] for k in out if k.startswith("xyz:")]
        for i, (n, _) in enumerate(self.fields):
            if n in out:
                fo.planes[i] = out[n].data_ptr()
            if "destaggered:" + n in out:
                fo.destaggered[i] = out["destaggered:" + n].data_ptr()
                any_dst = True
        for k, n in enumerate(xyz_names[:2]):
            t = out["xyz:" + n]
            fo.xyz[k] = t.data_ptr()
            fo.xyz_field[k] = self.field_index(n)
            fo.xyz_dtype = capi.F32 if t.dtype == torch.float32 else capi.F64
        for name in ("timestamp", "measurement_id", "status", "frame_meta", "packet_timestamp",
                     "alert_flags"):
            if name in out:
                setattr(fo, name, out[name].data_ptr())
        if gate is not None:
            lo, hi, empty = self.range_gate(*gate)
            if empty:
                lo, hi = 1, 0   # nothing passes
            if "gate_counts" not in out:
                out["gate_counts"] = torch.empty((n_frames, 8, self.w), dtype=torch.uint16, device="cuda")
            fo.gate_counts = out["gate_counts"].data_ptr()
            fo.gate_min_r, fo.gate_max_r = lo, hi
            fo.gate_field = self.field_index(gate_field)
        if poses is not None:
            assert poses.is_cuda and poses.dtype == torch.float64 and poses.is_contiguous()
            assert tuple(poses.shape) == (n_frames, self.w, 4, 4), poses.shape
            fo.xyz_poses = poses.data_ptr()
        shifts_p = None
        if any_dst:
            if self.shifts is None:
                raise ValueError("image height does not match shifts size")
            shifts_p = self.shifts.ctypes.data
        luts_arr = None
        n_luts = 0
        if xyz_names:
            n_luts = len(self.luts)
            luts_arr = (C.c_void_p * max(n_luts, 1))(*[l.h for l in self.luts])
        counts_p = None
        if isinstance(packet_counts, torch.Tensor):
            # device tensor (used in place; the graph-capturable form) or pinned host tensor
            assert packet_counts.dtype in (torch.uint32, torch.int32) and packet_counts.is_contiguous()
            assert packet_counts.numel() == n_frames
            counts_p = packet_counts.data_ptr()
        elif packet_counts is not None:
            packet_counts = np.ascontiguousarray(packet_counts, dtype=np.uint32)
            assert packet_counts.size == n_frames
            counts_p = packet_counts.ctypes.data
        capi.check(self.ctx.L.ouster_hip_decode(
            self.ctx.h, self.fmt.h, packets.data_ptr(), stride, slots, counts_p, n_frames,
            host_timestamps.data_ptr() if host_timestamps is not None else None, C.byref(fo),
            shifts_p, luts_arr, n_luts))

    def destagger(self, img: torch.Tensor, shifts=None, inverse: bool = False) -> torch.Tensor:
        """img: CUDA tensor [n, h, w, ...] or [h, w, ...]; returns the destaggered copy."""
        assert img.is_cuda and img.is_contiguous()
        sh = np.ascontiguousarray(self.shifts if shifts is None else shifts, dtype=np.int32)
        x = img
        if img.dim() == 2:
            n, h, w, extra = 1, img.shape[0], img.shape[1], 1
        else:
            n, h, w = img.shape[0], img.shape[1], img.shape[2]
            extra = int(np.prod(img.shape[3:])) if img.dim() > 3 else 1
        out = torch.empty_like(x)
        capi.check(self.ctx.L.ouster_hip_destagger(self.ctx.h, x.data_ptr(), out.data_ptr(), h, w,
                                                   img.element_size() * extra, sh.ctypes.data,
                                                   sh.size, int(inverse), n))
        return out

    def cartesian(self, rng: torch.Tensor, lut: Optional[capi.Lut] = None,
                  dtype=torch.float32) -> torch.Tensor:
        """rng: uint32 CUDA tensor [n, h, w] or [h, w] -> [n, h*w, 3]."""
        assert rng.is_cuda and rng.is_contiguous() and rng.dtype == torch.uint32
        lut = lut or self.luts[0]
        n = 1 if rng.dim() == 2 else rng.shape[0]
        if rng.numel() != n * self.h * self.w:
            raise ValueError("unexpected image dimensions")
        xyz = torch.empty((n, self.h * self.w, 3), dtype=dtype, device="cuda")
        capi.check(self.ctx.L.ouster_hip_cartesian(self.ctx.h, lut.h, rng.data_ptr(), xyz.data_ptr(),
                                                   capi.F32 if dtype == torch.float32 else capi.F64, n))
        return xyz

    def dewarp(self, points: torch.Tensor, poses: torch.Tensor) -> torch.Tensor:
        """points [n, h*w, 3] f32/f64, poses [n, w, 4, 4] f64 (per-column body_to_world)."""
        assert points.is_cuda and poses.is_cuda and poses.dtype == torch.float64
        assert points.is_contiguous() and poses.is_contiguous()
        n = points.shape[0]
        out = torch.empty_like(points)
        capi.check(self.ctx.L.ouster_hip_dewarp(
            self.ctx.h, points.data_ptr(), poses.data_ptr(), out.data_ptr(),
            capi.F32 if points.dtype == torch.float32 else capi.F64, self.h, self.w, n))
        return out

    def dewarp_frames(self, rng: torch.Tensor, status: torch.Tensor, poses: torch.Tensor,
                      min_range: float, max_range: float, timestamp: Optional[torch.Tensor] = None,
                      luts=None, dtype=torch.float32, provenance: bool = True,
                      capacity: Optional[int] = None,
                      gate_counts: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Range-gated, compacting dewarp of a batch of frames (impl/dewarp_impl.h:23-115).
        rng [n, h, w] u32, status [n, w] u32, poses [n, w, 4, 4] f64 -- or, for float output, [n, w, 3, 4] f32 = rows 0..2 of
        the poses already cast to float (pose_rows(): what dewarp<float> multiplies with anyway, 48 B per column instead of
        128) --, timestamp [n, w] u64.
        Returns {"points" [cap, 3], "frame_offsets" [n + 1] u64, and with provenance
        "frame_idxs", "col_idxs" (u32) and, when timestamp is given, "timestamps_ns" (u64)};
        the first frame_offsets[n] rows are valid."""
        assert rng.is_cuda and rng.is_contiguous() and rng.dtype == torch.uint32
        assert status.is_cuda and status.is_contiguous() and status.dtype == torch.uint32
        rows = poses.dtype == torch.float32
        assert poses.is_cuda and poses.is_contiguous() and (rows or poses.dtype == torch.float64)
        n = rng.shape[0]
        if rows and dtype != torch.float32:
            raise ValueError("float pose rows serve float output only")
        if rng.numel() != n * self.h * self.w or status.numel() != n * self.w or \
                poses.numel() != n * self.w * (12 if rows else 16):
            raise ValueError("unexpected image dimensions")
        luts = list(luts) if luts is not None else self.luts
        cap = n * self.h * self.w if capacity is None else int(capacity)
        out = {"points": torch.empty((cap, 3), dtype=dtype, device="cuda"),
               "frame_offsets": torch.empty(n + 1, dtype=torch.uint64, device="cuda")}
        if provenance:
            out["frame_idxs"] = torch.empty(cap, dtype=torch.uint32, device="cuda")
            out["col_idxs"] = torch.empty(cap, dtype=torch.uint32, device="cuda")
            if timestamp is not None:
                assert timestamp.is_cuda and timestamp.is_contiguous() and timestamp.dtype == torch.uint64
                out["timestamps_ns"] = torch.empty(cap, dtype=torch.uint64, device="cuda")
        luts_arr = (C.c_void_p * max(len(luts), 1))(*[l.h for l in luts])
        ptr = lambda k: out[k].data_ptr() if k in out else None
        if gate_counts is not None:
            assert gate_counts.is_cuda and gate_counts.dtype == torch.uint16 and gate_counts.is_contiguous()
            assert tuple(gate_counts.shape) == (n, 8, self.w)
        if rows:
            capi.check(self.ctx.L.ouster_hip_dewarp_frames_rows(
                self.ctx.h, luts_arr, len(luts), rng.data_ptr(), status.data_ptr(),
                timestamp.data_ptr() if timestamp is not None else None, poses.data_ptr(), n,
                float(min_range), float(max_range), out["points"].data_ptr(), ptr("frame_idxs"), ptr("col_idxs"),
                ptr("timestamps_ns"), cap, out["frame_offsets"].data_ptr(),
                gate_counts.data_ptr() if gate_counts is not None else None))
            return out
        capi.check(self.ctx.L.ouster_hip_dewarp_frames_counted(
            self.ctx.h, luts_arr, len(luts), rng.data_ptr(), status.data_ptr(),
            timestamp.data_ptr() if timestamp is not None else None, poses.data_ptr(), n,
            float(min_range), float(max_range), capi.F32 if dtype == torch.float32 else capi.F64,
            out["points"].data_ptr(), ptr("frame_idxs"), ptr("col_idxs"), ptr("timestamps_ns"),
            cap, out["frame_offsets"].data_ptr(),
            gate_counts.data_ptr() if gate_counts is not None else None))
        return out

    @staticmethod
    def pose_rows(poses) -> torch.Tensor:
        """[n, w, 4, 4] float64 poses (host array or tensor, any device) -> [n, w, 3, 4] float32 CUDA tensor: rows 0..2 cast
        to float, the form dewarp<float> uses (pose_util.h:38-56)."""
        t = torch.as_tensor(poses)
        return t[..., :3, :].to(torch.float32).contiguous().cuda()

    def sync(self):
        self.ctx.sync()
