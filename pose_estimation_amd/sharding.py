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
    # rank r's row k is item k*world + r
    full = torch.empty((n_items, W), dtype=local.dtype, device=local.device)
    for r in range(world):
        n_r = shard_size(n_items, r, world)
        full[r::world][:n_r] = out[r * per: r * per + n_r]
    return full
