"""Multi-GPU plumbing: one process per GPU, independent triplets sharded across ranks, weights broadcast once.

The reference's "multi-GPU" path is a multiprocessing.Pool with one worker per GPU id, each job re-loading the
298 MB checkpoint from disk (eval/generate_inpaintinggame_wb_saliency_maps_multigpu.py:74,121-231).  Here rank 0
packs the parameters once and the packed arena (W, relu(W), backward-packed relu(W), folded BatchNorm) is sent to the
other ranks with ONE collective -- torch.distributed.broadcast, i.e. RCCL over xGMI with backend "nccl" (gloo in the
CPU tests).  The steady state has no collective at all: triplets are independent.
"""
import os

import torch
import torch.distributed as dist


def dist_env():
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    if 'XFR_FORCE_DEVICE' in os.environ:      # test hook: several ranks on one GPU (gloo)
        local = int(os.environ['XFR_FORCE_DEVICE'])
    return rank, world, local


def init_process_group(backend=None, force=False):
    """force: initialise the group even at world size 1 (exercises the RCCL code path on a single GPU)."""
    rank, world, local = dist_env()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if backend is None:
            backend = os.environ.get('XFR_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for `rank` (sizes differ by at most one)."""
    q, r = divmod(int(n_items), int(world))
    lo = rank * q + min(rank, r)
    hi = lo + q + (1 if rank < r else 0)
    return lo, hi


def broadcast_arena(arena, src=0):
    """Broadcast a packed parameter arena (uint8 tensor; CUDA for RCCL, CPU for gloo) from rank `src`."""
    if dist.is_initialized():
        dist.broadcast(arena, src=src)
    return arena


def load_and_broadcast(engine, state_dict_fn, src=0):
    """Rank `src` packs the parameters (state_dict_fn() is only called there); everyone receives the arena."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank == src:
        engine.load_weights(state_dict_fn())
    arena = engine.weight_arena()
    broadcast_arena(arena, src)
    if rank != src:
        engine.mark_weights_loaded()
    return engine


def gather_maps(local_maps, n_total):
    """All-gather per-rank saliency maps [n_local, H, W] into [n_total, H, W] on every rank (optional; ranks can
    equally write their shard to disk, like the reference's workers do)."""
    if not dist.is_initialized():
        return local_maps
    world = dist.get_world_size()
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local_maps.shape[1:]), dtype=local_maps.dtype, device=local_maps.device)
    pad[:local_maps.shape[0]] = local_maps
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[:hi - lo] for o, (lo, hi) in zip(outs, sizes)], dim=0)
