"""Sharding of independent pairs / frames over ranks, and the single final gather.

One process per GPU.  Every rank holds all per-view tables (they fit 288 GB many times over) and
processes the pairs `pair_id % world == rank` (block-cyclic, balances per-view reuse).  There is
no data-path collective; the only exchange is one all_gather of the [pairs_local, 16] float32
pose table (RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world: int) -> torch.Tensor:
    """Item ids owned by `rank` (int64, ascending)."""
    if rank >= n_items:
        return torch.empty(0, dtype=torch.int64)
    return torch.arange(rank, n_items, world, dtype=torch.int64)


def shard_size(n_items: int, rank: int, world: int) -> int:
    return (n_items - rank + world - 1) // world if n_items > rank else 0


def gather_pose_table(local: torch.Tensor, n_items: int, group=None) -> torch.Tensor:
    """All ranks' [n_local, W] tables -> the full [n_items, W] table in item order, on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        return local
    W = local.shape[1]
    per = (n_items + world - 1) // world                      # rank 0 holds the most items
    padded = torch.zeros((per, W), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world * per, W), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return interleave_shards(out, n_items, world)


def interleave_shards(stacked: torch.Tensor, n_items: int, world: int) -> torch.Tensor:
    """[world * per, W] (rank r's rows, zero-padded to per = ceil(n_items / world), one rank after the other — the
    layout all_gather_into_tensor leaves) -> [n_items, W] in item order: rank r's row k is item k * world + r."""
    per = (n_items + world - 1) // world
    full = torch.empty((n_items, stacked.shape[1]), dtype=stacked.dtype, device=stacked.device)
    for r in range(world):
        n_r = shard_size(n_items, r, world)
        full[r::world][:n_r] = stacked[r * per: r * per + n_r]
    return full


def run_pair_list(vi, vj, process_chunk, rank: int, world: int, chunk: int = 64, stream_ctx=None, sync=None,
                  progress=None, group=None, width: int = 16, device=None):
    """The all-pairs runner of cfg 3/4 (the reference's O(n^2) pair loop, main.py:458-463, sharded and batched).

    vi, vj: the FULL lexicographic pair list (array-likes of equal length); this rank takes the pairs
    ``pair_id % world == rank``, hands them to ``process_chunk(k, ids, vi_chunk, vj_chunk) -> [len, W] table`` in
    chunks of `chunk` pairs (``ids`` = global pair ids of the chunk, int64 tensor; chunk k may be enqueued inside
    ``stream_ctx(k)``, a context manager — HIP streams on a GPU, nothing on CPU), calls ``sync()`` once after the
    last chunk and then does the ONE collective of the job: `gather_pose_table` of the rank's rows.

    Returns (full [n_pairs, W] table in pair order — on every rank —, local [n_local, W] table, local pair ids)."""
    import contextlib
    n_items = len(vi)
    mine = shard_indices(n_items, rank, world)
    vi_l, vj_l = vi[mine.numpy()], vj[mine.numpy()]
    tables = []
    for k, s in enumerate(range(0, len(mine), chunk)):
        e = min(s + chunk, len(mine))
        with (stream_ctx(k) if stream_ctx is not None else contextlib.nullcontext()):
            tables.append(process_chunk(k, mine[s:e], vi_l[s:e], vj_l[s:e]))
        if progress is not None:
            progress(k, e)
    if sync is not None:
        sync()
    # a rank beyond the end of the pair list owns nothing but still takes part in the gather
    local = torch.cat(tables) if tables else torch.zeros((0, width), dtype=torch.float32, device=device or "cpu")
    full = gather_pose_table(local, n_items, group=group)
    return full, local, mine
