"""N > 1 path on CPU: world_size-2 gloo processes shard a pair list and gather the pose table."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pose_estimation_amd import sharding


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = sharding.shard_indices(n_items, rank, world)
    assert len(ids) == sharding.shard_size(n_items, rank, world)
    local = torch.stack([ids.float() * 10 + c for c in range(16)], dim=1) if len(ids) else torch.zeros((0, 16))
    full = sharding.gather_pose_table(local, n_items)
    q.put((rank, full.numpy().copy()))   # by value: a shared-memory tensor handle dies with the sender
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 8, 1])
def test_shard_and_gather_world2(n_items):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = torch.stack([torch.arange(n_items).float() * 10 + c for c in range(16)], dim=1)
    for r in range(world):
        assert torch.equal(torch.from_numpy(got[r]), want)


def _runner_worker(rank, world, port, n_views, chunk, q):
    """The runner of cfg 3/4 (`sharding.run_pair_list`: shard, chunk, sync, ONE gather) with a CPU stand-in for the
    GPU chunk processor: a row is a pure function of the global pair id and its two views."""
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    i, j = np.triu_indices(n_views, k=1)
    vi, vj = i.astype(np.int32), j.astype(np.int32)
    calls = []

    def chunk_fn(k, ids, a, b):
        assert len(ids) == len(a) == len(b) <= chunk
        calls.append(len(ids))
        cols = [ids.float(), torch.from_numpy(a.copy()).float(), torch.from_numpy(b.copy()).float()]
        return torch.stack(cols + [ids.float() * 0.5 + c for c in range(13)], dim=1)

    synced = []
    full, local, mine = sharding.run_pair_list(vi, vj, chunk_fn, rank, world, chunk=chunk, sync=lambda: synced.append(1))
    assert synced == [1] and sum(calls) == len(mine) == sharding.shard_size(len(vi), rank, world)
    assert torch.equal(mine, sharding.shard_indices(len(vi), rank, world)) and local.shape == (len(mine), 16)
    q.put((rank, full.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_views,chunk", [(12, 8), (3, 64), (2, 4)])
def test_runner_shard_and_final_gather_world2(n_views, chunk):
    import numpy as np
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_runner_worker, args=(r, world, port, n_views, chunk, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    i, j = np.triu_indices(n_views, k=1)
    ids = np.arange(len(i), dtype=np.float32)
    want = np.stack([ids, i.astype(np.float32), j.astype(np.float32)] + [ids * 0.5 + c for c in range(13)], axis=1)
    for r in range(world):
        assert np.array_equal(got[r], want)


def test_interleave_shards_matches_the_gather_layout():
    for n, w in [(7, 2), (780, 2), (5, 8), (16, 4)]:
        per = (n + w - 1) // w
        stacked = torch.zeros((w * per, 3))
        for r in range(w):
            ids = sharding.shard_indices(n, r, w)
            stacked[r * per: r * per + len(ids), 0] = ids.float()
        assert torch.equal(sharding.interleave_shards(stacked, n, w)[:, 0], torch.arange(n).float())


def test_shard_indices_cover_everything_once():
    for n, w in [(4950, 8), (10, 3), (0, 2), (5, 8)]:
        seen = torch.cat([sharding.shard_indices(n, r, w) for r in range(w)])
        assert sorted(seen.tolist()) == list(range(n))


def test_gather_without_process_group_is_identity():
    t = torch.ones((3, 16))
    assert sharding.gather_pose_table(t, 3) is t
