"""Multi-process path on CPU (gloo, world_size 2): shard partition, one-off arena broadcast, map gather."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from xfr_amd import shard


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 32, 541):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, l = shard.init_process_group('gloo')
    assert (r, w) == (rank, world)
    # rank 0 "packs" the arena, the other rank receives it (stand-in for the 490 MB packed ResNet-101 parameters)
    arena = torch.arange(4096, dtype=torch.uint8) if rank == 0 else torch.zeros(4096, dtype=torch.uint8)
    shard.broadcast_arena(arena, src=0)
    ok_b = bool((arena == torch.arange(4096, dtype=torch.uint8)).all())
    # independent units: each rank produces the maps of its shard, gathered for the check
    n_total = 7
    lo, hi = shard.shard_range(n_total, rank, world)
    local = torch.stack([torch.full((4, 4), float(i)) for i in range(lo, hi)]) if hi > lo else torch.zeros((0, 4, 4))
    allm = shard.gather_maps(local, n_total)
    ok_g = bool((allm[:, 0, 0] == torch.arange(n_total, dtype=torch.float32)).all())
    q.put((rank, ok_b, ok_g))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world_size_2_broadcast_and_gather():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1] and all(r[1] and r[2] for r in res)
