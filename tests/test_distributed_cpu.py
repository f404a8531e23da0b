"""N > 1 path on CPU: world_size-2 gloo processes exercise the frame sharding, the
max-over-ranks timing reduction and the optional scatter/gather exchange that bench.py and
multi-GPU callers use (ouster_sdk_amd/parallel.py).  No GPU, no kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ouster_sdk_amd import parallel


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 511, 512):
        for world in (1, 2, 3, 8):
            blocks = [parallel.shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for (b0, e0), (b1, e1) in zip(blocks, blocks[1:]):
                assert e0 == b1
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_range(4, 2, 2)


def test_sensor_tick_owner_covers_world():
    for world in (1, 2, 4, 8):
        owners = {parallel.sensor_tick_owner(s, t, 4, world) for s in range(4) for t in range(16)}
        assert owners == set(range(world)) or world > 4 and len(owners) >= 4
        # a sensor never bounces across more than world/4 GPUs
        for s in range(4):
            own = {parallel.sensor_tick_owner(s, t, 4, world) for t in range(16)}
            assert len(own) <= max(1, world // 4)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, lr, w = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world)
    frame_shape = (4, 33)
    batch = None
    if rank == 0:
        batch = torch.arange(n_frames * 4 * 33, dtype=torch.int32).reshape(n_frames, *frame_shape)
    mine = parallel.scatter_frames(batch, n_frames, frame_shape, torch.int32, "cpu")
    b, e = parallel.shard_range(n_frames, rank, world)
    want = torch.arange(n_frames * 4 * 33, dtype=torch.int32).reshape(n_frames, *frame_shape)[b:e]
    assert torch.equal(mine, want)
    out = parallel.gather_frames(mine * 2, n_frames)          # "process" = x2, frames independent
    if rank == 0:
        assert torch.equal(out, want.new_tensor(
            np.arange(n_frames * 4 * 33, dtype=np.int32).reshape(n_frames, *frame_shape) * 2))
    else:
        assert out is None
    t = parallel.max_over_ranks(1.0 + rank)
    assert t == float(world)
    parallel.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 8])
def test_gloo_world2_scatter_process_gather(n_frames):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_frames), nprocs=2, join=True)
