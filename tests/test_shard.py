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
    # per-rank rates: rank 1 plays the straggler (half the rate); every rank sees it in `spread`
    rr = shard.gather_rank_rates(100.0 if rank == 0 else 50.0)
    ok_r = rr['per_rank'] == [100.0, 50.0] and rr['min'] == 50.0 and rr['max'] == 100.0 and abs(rr['spread'] - 0.5) < 1e-12
    q.put((rank, ok_b, ok_g and ok_r))
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


def test_rank_rates_single_process():
    rr = shard.gather_rank_rates(1234.5)
    assert rr == {'per_rank': [1234.5], 'min': 1234.5, 'max': 1234.5, 'spread': 0.0}


def test_normalize_gpus_like_the_reference():
    # no mask: unchanged (utils.py:521-525)
    assert shard.normalize_gpus([0, 1], env={}) == [0, 1]
    assert shard.normalize_gpus([0, 1], env={'HIP_VISIBLE_DEVICES': ''}) == [0, 1]
    # logical index -> entry of the mask (utils.py:531-536); HIP's mask wins over CUDA's
    assert shard.normalize_gpus([0, 2], env={'HIP_VISIBLE_DEVICES': '4,5,7'}) == [4, 7]
    assert shard.normalize_gpus([1], env={'CUDA_VISIBLE_DEVICES': '3,2'}) == [2]
    assert shard.normalize_gpus([1], env={'HIP_VISIBLE_DEVICES': '6,1', 'CUDA_VISIBLE_DEVICES': '3,2'}) == [1]
    assert shard.normalize_gpus([0], env={'ROCR_VISIBLE_DEVICES': 'GPU-abcdef'}) == ['GPU-abcdef']
    # both levels set: HIP indices refer to the ROCR-filtered list
    assert shard.normalize_gpus([0, 1], env={'ROCR_VISIBLE_DEVICES': '4,5,6,7', 'HIP_VISIBLE_DEVICES': '2,0'}) == [6, 4]
    with pytest.raises(ValueError):
        shard.normalize_gpus([0], env={'ROCR_VISIBLE_DEVICES': '4,5', 'HIP_VISIBLE_DEVICES': '3'})
    # fewer CPUs than local ranks: a report, not an exception
    rep = shard.bind_rank_cpus(0, 4, nodes=[None] * 4, allowed=[0, 1])
    assert rep['cpus'] is None and rep['how'].startswith('not bound')
    # more GPUs than visible / index outside the range: ValueError (utils.py:528-529, 535-536)
    with pytest.raises(ValueError):
        shard.normalize_gpus([0, 1, 2], env={'HIP_VISIBLE_DEVICES': '0,1'})
    with pytest.raises(ValueError):
        shard.normalize_gpus([5], env={'HIP_VISIBLE_DEVICES': '0,1'})


def test_cpu_binding_slices():
    assert shard.parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    sl = shard.cpu_slices(range(16), 3)
    assert sum(sl, []) == list(range(16)) and max(map(len, sl)) - min(map(len, sl)) <= 1
    with pytest.raises(ValueError):
        shard.cpu_slices([0, 1], 3)
    # no NUMA information (nodes unknown): an even split of the allowed CPUs; the calling process really is pinned, then restored
    before = os.sched_getaffinity(0)
    allowed = sorted(before)
    if len(allowed) >= 2:
        try:
            b0 = shard.bind_rank_cpus(0, 2, nodes=[None, None], allowed=allowed)
            got0 = sorted(os.sched_getaffinity(0))
            os.sched_setaffinity(0, before)
            b1 = shard.bind_rank_cpus(1, 2, nodes=[None, None], allowed=allowed)
            got1 = sorted(os.sched_getaffinity(0))
        finally:
            os.sched_setaffinity(0, before)
        assert got0 + got1 == allowed and not set(got0) & set(got1)
        assert b0['n_cpus'] == len(got0) and b1['n_cpus'] == len(got1) and 'even split' in b0['how']


class _FakeEngine(object):
    """load_weights packs deterministically from the state dict; weight_arena is the packed tensor (the engine's contract, on the CPU)."""

    def __init__(self):
        self.arena = torch.zeros(1024, dtype=torch.uint8)
        self.packed = self.marked = 0

    def load_weights(self, sd):
        self.arena.copy_(sd['w'])
        self.packed += 1

    def weight_arena(self):
        return self.arena

    def mark_weights_loaded(self):
        self.marked += 1


def _comm_worker(rank, world, port, q, scenario):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      XFR_DIST_BACKEND='gloo', XFR_DIST_TIMEOUT='30')
    if scenario == 'fail_broadcast':
        os.environ['XFR_TEST_FAIL_BROADCAST'] = '1'
    if scenario == 'fail_init':
        os.environ['XFR_TEST_FAIL_INIT'] = '1' if rank == 1 else '0'
    comm = shard.Comm()
    eng = _FakeEngine()
    want = (torch.arange(1024) % 251).to(torch.uint8)
    comm.load_weights(eng, lambda: {'w': want})
    out = {'rank': rank, 'via': comm.weights_via, 'arena_ok': bool((eng.arena == want).all()), 'packed': eng.packed, 'collective_ok': comm.collective_ok}
    out['reports'] = comm.gather_objects({'r': rank}, 'rep')
    out['max'] = comm.max_float(10.0 + rank)
    comm.barrier()
    if scenario == 'raise':
        # rank 1 fails between two barriers: rank 0's next barrier raises with rank 1's text instead of hanging
        if rank == 1:
            comm.report_error('Traceback ...\nRuntimeError: boom on rank 1')
            q.put(out)
            return
        try:
            comm.barrier()
            out['barrier_raised'] = None
        except RuntimeError as ex:
            out['barrier_raised'] = str(ex)
        out['errors'] = comm.collect_errors()
        q.put(out)
        return
    q.put(out)
    comm.close()


@pytest.mark.parametrize('scenario', ['healthy', 'fail_broadcast', 'fail_init', 'raise'])
def test_comm_survives_a_failing_collective_backend(scenario):
    """shard.Comm (what bench.py --gpus N runs on): healthy -> ONE broadcast, rank 1 never packs; a broadcast that fails on one rank or a
    collective backend that does not come up on one rank -> EVERY rank packs locally ('local_pack_fallback'), reports / max / barrier go through
    the store; a rank that dies between barriers turns the others' next barrier into an error that carries its text.  (Every rank must also EXIT cleanly:
    without a collective backend nothing but Comm.close's check-out keeps rank 0 -- whose process holds the store -- alive until the other rank has left
    its last store barrier; before round 6's check-out this scenario failed every other run with 'Connection was likely closed'.)"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29950 + (os.getpid() % 200) + {'healthy': 0, 'fail_broadcast': 211, 'fail_init': 422, 'raise': 633}[scenario]
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, q, scenario)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda r: r['rank'])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r['arena_ok'] for r in res)
    assert all(r['reports'] == [{'r': 0}, {'r': 1}] and r['max'] == 11.0 for r in res)
    if scenario in ('healthy', 'raise'):
        assert [r['via'] for r in res] == ['broadcast', 'broadcast'] and [r['packed'] for r in res] == [1, 0]
    else:
        assert [r['via'] for r in res] == ['local_pack_fallback'] * 2 and all(r['packed'] >= 1 for r in res)
    if scenario == 'fail_init':
        assert not any(r['collective_ok'] for r in res)
    if scenario == 'raise':
        assert 'boom on rank 1' in res[0]['barrier_raised'] and 1 in res[0]['errors'] or '1' in res[0]['errors']

