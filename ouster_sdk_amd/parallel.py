"""Frame sharding across GPUs (one process per GPU, torch.distributed; RCCL on ROCm).

Frames (and sensors) are independent units of the hot path -- no halo, no reduction -- so
ranks shard a batch of frames with NO collective on the data path.  The only real exchange
step is optional: when a batch originates on one rank, `scatter_frames` hands every rank its
contiguous block of raw packet buffers and `gather_frames` collects results (XYZ) back on
the root.  Both work on any backend ("nccl" = RCCL over xGMI on GPUs, "gloo" on CPU for
tests).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [begin, end) of `n_items` owned by `rank` (sizes differ by <= 1)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def sensor_tick_owner(sensor: int, tick: int, n_sensors: int, world: int) -> int:
    """Round-robin (sensor, tick) placement for multi-sensor streams: a sensor's frames stay
    on the same GPUs so its LUT tables stay hot."""
    per_sensor = max(1, world // n_sensors)
    return (sensor % world) if world <= n_sensors else \
        (sensor * per_sensor + tick % per_sensor) % world


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from torchrun's environment; initialises the process group
    when world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _p2p(ops):
    """Run a list of point-to-point ops as ONE group (ncclGroupStart/End on RCCL: the root's sends to
    its 7 peers go out concurrently, one xGMI link each) and wait for all of them."""
    if not ops:
        return
    for req in dist.batch_isend_irecv(ops):
        req.wait()


def scatter_frames(batch: Optional[torch.Tensor], n_frames: int, frame_shape, dtype, device,
                   src: int = 0) -> torch.Tensor:
    """Root holds `batch` [n_frames, *frame_shape]; every rank receives its shard_range block.
    Point-to-point, grouped: blocks may differ in size by one frame, and nothing but the shard a rank
    owns ever travels to it."""
    rank, world = dist.get_rank(), dist.get_world_size()
    b, e = shard_range(n_frames, rank, world)
    mine = torch.empty((e - b, *frame_shape), dtype=dtype, device=device)
    ops = []
    if rank == src:
        batch = batch.contiguous()   # RCCL sends need dense memory; a no-op for the usual dense batch
        for r in range(world):
            rb, re = shard_range(n_frames, r, world)
            if r == src:
                mine.copy_(batch[rb:re])
            elif re > rb:
                ops.append(dist.P2POp(dist.isend, batch[rb:re], r))   # a contiguous slice of the batch
    elif e > b:
        ops.append(dist.P2POp(dist.irecv, mine, src))
    _p2p(ops)
    return mine


def gather_frames(mine: torch.Tensor, n_frames: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Inverse of scatter_frames: root returns [n_frames, ...], others None."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == dst:
        out = torch.empty((n_frames, *mine.shape[1:]), dtype=mine.dtype, device=mine.device)
        ops = []
        for r in range(world):
            rb, re = shard_range(n_frames, r, world)
            if r == dst:
                out[rb:re].copy_(mine)
            elif re > rb:
                ops.append(dist.P2POp(dist.irecv, out[rb:re], r))
        _p2p(ops)
        return out
    if mine.shape[0] > 0:
        _p2p([dist.P2POp(dist.isend, mine.contiguous(), dst)])
    return None
