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
    part = torch.full((H, W, 4), float(rank + 1))
    total = rdist.reduce_frame(part, rank, world)
    if rank == 0:
        q.put((frame.numpy(), total.numpy()))
    else:
        assert frame is None and total is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_and_reduce():
    H, W, tile_rows, world = 37, 5, 4, 2
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
    assert np.all(total == 3.0)
