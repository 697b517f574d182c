"""The N > 1 assembly path on CPU: two gloo ranks, fake shards whose pixels encode their global
row, gathered and un-interleaved by raytrace_amd.dist exactly as bench.py does with RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, H, W, tile_rows, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytrace_amd import dist as rdist
    rows = rdist.shard_row_map(H, tile_rows, rank, world)
    shard = torch.full((rdist.max_shard_rows(H, tile_rows, world), W, 4), -1.0)
    shard[: len(rows)] = torch.as_tensor(rows, dtype=torch.float32)[:, None, None] * 10 + rank
    frame = rdist.gather_frame(shard, H, tile_rows, rank, world)
    # sample-split: integer images (32.32 fixed-point sums, sample count) of this rank's window
    begin, count = rdist.sample_window(100, rank, world)
    part = torch.zeros((H, W, 4), dtype=torch.int64)
    part[..., :3] = (torch.arange(H * W * 3, dtype=torch.int64).reshape(H, W, 3) * 7919 + (1 << 40)) * count + rank
    part[..., 3] = count
    total = rdist.reduce_sums(part, rank, world)
    if rank == 0:
        q.put((frame.numpy(), total.numpy()))
    else:
        assert frame is None and total is None
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_gather_and_reduce(world):
    H, W, tile_rows = 37, 5, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, tile_rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    frame, total = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    owner = (np.arange(H) // tile_rows) % world
    want = np.arange(H) * 10 + owner
    assert np.array_equal(frame[:, 0, 0], want.astype(np.float32))
    assert np.all(frame == frame[:, :1, :1])
    base = np.arange(H * W * 3, dtype=np.int64).reshape(H, W, 3) * 7919 + (1 << 40)
    assert np.array_equal(total[..., :3], base * 100 + sum(range(world)))       # exact, whatever the order
    assert np.all(total[..., 3] == 100)


def test_sample_windows_and_gather_index_cover_everything():
    from raytrace_amd import dist as rdist
    for spp, world in ((256, 8), (100, 3), (5, 8), (1000, 7)):
        w = [rdist.sample_window(spp, g, world) for g in range(world)]
        assert w[0][0] == 0 and sum(c for _, c in w) == spp
        assert all(w[g][0] + w[g][1] == w[g + 1][0] for g in range(world - 1))
        assert max(c for _, c in w) - min(c for _, c in w) <= 1
    for H, tr, world in ((1080, 4, 8), (37, 4, 3), (10, 3, 2), (2160, 4, 5)):
        mr = rdist.max_shard_rows(H, tr, world)
        idx = rdist.gather_index(H, tr, world)
        assert len(np.unique(idx)) == H
        for g in range(world):
            rows = rdist.shard_row_map(H, tr, g, world)
            assert np.array_equal(idx[rows], g * mr + np.arange(len(rows)))
