"""Row-tile sharding of one frame over the GPUs of a node (one process per GPU) and the gather
of the partial framebuffers over RCCL (torch.distributed backend "nccl"; "gloo" in CPU tests).

Rows are cut into tiles of `tile_rows` rows; rank g renders tiles g, g+N, g+2N, ... and packs
their rows in that order (the tile_rows/tile_first/tile_step fields of rt_render_params).
Interleaving keeps the ranks balanced although the lower half of a frame traces several times
more rays than the sky.  Every (pixel, sample) draws from its own Philox key, so the assembled
frame is bit-identical to the single-GPU frame.
"""
from __future__ import annotations

import numpy as np


def shard_row_map(height: int, tile_rows: int, rank: int, world: int) -> np.ndarray:
    """Global row index of every packed row of `rank`'s shard."""
    n_tiles = (height + tile_rows - 1) // tile_rows
    rows = []
    for t in range(rank, n_tiles, world):
        rows.extend(range(t * tile_rows, min((t + 1) * tile_rows, height)))
    return np.asarray(rows, dtype=np.int64)


def max_shard_rows(height: int, tile_rows: int, world: int) -> int:
    return max(len(shard_row_map(height, tile_rows, g, world)) for g in range(world))


_ROW_INDEX_CACHE = {}


def _row_index(height, tile_rows, g, world, device):
    import torch
    key = (height, tile_rows, g, world, str(device))
    if key not in _ROW_INDEX_CACHE:
        _ROW_INDEX_CACHE[key] = torch.as_tensor(shard_row_map(height, tile_rows, g, world), device=device)
    return _ROW_INDEX_CACHE[key]


def gather_frame(shard, height: int, tile_rows: int, rank: int, world: int, dst: int = 0, group=None):
    """shard: torch tensor [max_shard_rows, W, C] (rows past this rank's count are padding).
    Returns the assembled [height, W, C] frame on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return shard[:height]
    try:
        bufs = [torch.empty_like(shard) for _ in range(world)] if rank == dst else None
        dist.gather(shard, bufs, dst=dst, group=group)
    except (RuntimeError, NotImplementedError):
        # a backend without gather: every rank receives every shard instead
        flat = torch.empty((world,) + tuple(shard.shape), dtype=shard.dtype, device=shard.device)
        dist.all_gather_into_tensor(flat, shard, group=group)
        bufs = list(flat.unbind(0))
    if rank != dst:
        return None
    frame = torch.empty((height,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
    for g in range(world):
        rows = _row_index(height, tile_rows, g, world, shard.device)
        frame[rows] = bufs[g][: rows.numel()]
    return frame


def reduce_frame(partial, rank: int, world: int, dst: int = 0, group=None):
    """Sample-split: every rank rendered the whole frame for its own sample range
    (sample_begin = rank * spp); the float4 sums add up on `dst`."""
    import torch.distributed as dist

    if world > 1:
        dist.reduce(partial, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return partial if rank == dst else None
