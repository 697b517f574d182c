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


_GATHER_INDEX_CACHE = {}


def gather_index(height: int, tile_rows: int, world: int, max_rows: int | None = None) -> np.ndarray:
    """For every global row, its position in the ranks' packed shards laid end to end
    (rank g's shard starts at g * max_rows): the whole un-interleave is one index_select."""
    if max_rows is None:
        max_rows = max_shard_rows(height, tile_rows, world)
    index = np.empty(height, dtype=np.int64)
    for g in range(world):
        rows = shard_row_map(height, tile_rows, g, world)
        index[rows] = g * max_rows + np.arange(len(rows))
    return index


def _gather_index(height, tile_rows, world, max_rows, device):
    import torch
    key = (height, tile_rows, world, max_rows, str(device))
    if key not in _GATHER_INDEX_CACHE:
        _GATHER_INDEX_CACHE[key] = torch.as_tensor(gather_index(height, tile_rows, world, max_rows), device=device)
    return _GATHER_INDEX_CACHE[key]


def gather_frame(shard, height: int, tile_rows: int, rank: int, world: int, dst: int = 0, group=None, always=False):
    """shard: torch tensor [max_shard_rows, W, C] (rows past this rank's count are padding), the
    same shape on every rank.  Returns the assembled [height, W, C] frame on `dst`, None elsewhere.
    The shards arrive side by side in one buffer and one precomputed row index puts every row in
    its place.  always: go through the collective and the index among one rank too (rehearsals of
    the RCCL path on one GPU)."""
    import torch
    import torch.distributed as dist

    if world == 1 and not always:
        return shard[:height]
    flat = torch.empty((world,) + tuple(shard.shape), dtype=shard.dtype, device=shard.device) if rank == dst else None
    dist.gather(shard, list(flat.unbind(0)) if rank == dst else None, dst=dst, group=group)
    if rank != dst:
        return None
    index = _gather_index(height, tile_rows, world, shard.shape[0], shard.device)
    return flat.view((world * shard.shape[0],) + tuple(shard.shape[1:])).index_select(0, index)


def reduce_sums(partial, rank: int, world: int, dst: int = 0, group=None, always=False):
    """Sample-split: every rank rendered the whole frame for its own sample window (sample_begin)
    with RT_FLAG_FIXED_SUMS, `partial` holding the [H, W, 4] integer image as int64 (r, g, b sums in
    32.32 fixed point, sample count).  Integer addition is exact and order-free, so the sum on
    `dst` is bit for bit what one GPU accumulates for all the samples, whatever the reduction
    tree; rt_resolve_sums turns it into the float frame.  Reduces in place."""
    import torch
    import torch.distributed as dist

    if partial.dtype != torch.int64:
        raise TypeError("reduce_sums takes the RT_FLAG_FIXED_SUMS image as int64")
    if world > 1 or always:
        dist.reduce(partial, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return partial if rank == dst else None


def sample_window(spp: int, rank: int, world: int):
    """(sample_begin, count) of `rank`'s share of `spp` samples: windows differ by at most one."""
    base, extra = divmod(spp, world)
    begin = rank * base + min(rank, extra)
    return begin, base + (1 if rank < extra else 0)
