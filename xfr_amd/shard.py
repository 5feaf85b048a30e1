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


def normalize_gpus(gpus, env=None):
    """Logical GPU indices (command line, LOCAL_RANK) -> the ids the runtime was told to expose, like the reference's
    `normalize_gpus` does with CUDA_VISIBLE_DEVICES (python/xfr/utils.py:515-540).  ROCm composes its masks: ROCR_VISIBLE_DEVICES filters
    the agents the HIP runtime sees, HIP_VISIBLE_DEVICES (or, when that is unset, CUDA_VISIBLE_DEVICES) then indexes the FILTERED list --
    so a logical index goes through the HIP-level mask first and its result through the ROCR mask.  No mask set: the list is returned
    unchanged.  Raises ValueError like the reference when an index lies outside a visible range."""
    env = os.environ if env is None else env

    def through(ids, mask, name):
        visible = [v.strip() for v in mask.split(',')]
        if len(visible) < len(ids):
            raise ValueError('more GPUs requested than are visible through %s' % name)
        out = []
        for g in ids:
            if not str(g).lstrip('-').isdigit() or not 0 <= int(g) < len(visible):
                raise ValueError('GPU %s is outside the range visible through %s' % (g, name))
            v = visible[int(g)]
            out.append(int(v) if v.lstrip('-').isdigit() else v)
        return out

    out = list(gpus)
    hip_name = 'HIP_VISIBLE_DEVICES' if env.get('HIP_VISIBLE_DEVICES') else 'CUDA_VISIBLE_DEVICES'
    if env.get(hip_name):
        out = through(out, env[hip_name], hip_name)
    if env.get('ROCR_VISIBLE_DEVICES'):
        out = through(out, env['ROCR_VISIBLE_DEVICES'], 'ROCR_VISIBLE_DEVICES')
    return out


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (sysfs cpulist format)."""
    cpus = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def cpu_slices(cpus, n_ranks):
    """Split a CPU list into n_ranks contiguous, disjoint, non-empty slices (sizes differ by at most one)."""
    cpus = sorted(cpus)
    if n_ranks < 1 or len(cpus) < n_ranks:
        raise ValueError('%d CPUs cannot be split among %d ranks' % (len(cpus), n_ranks))
    return [cpus[slice(*shard_range(len(cpus), r, n_ranks))] for r in range(n_ranks)]


def gpu_numa_node(local):
    """NUMA node of logical device `local` from sysfs (PCI address from the device properties), or None."""
    try:
        p = torch.cuda.get_device_properties(local)
        bdf = '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open('/sys/bus/pci/devices/%s/numa_node' % bdf).read())
        return node if node >= 0 else None
    except Exception:
        return None


def bind_rank_cpus(local, local_world, nodes=None, allowed=None):
    """One CPU set per rank: the CPUs of the GPU's NUMA node, split among the ranks whose GPUs sit on that node (the reference's
    workers are unpinned, generate_inpaintinggame_wb_saliency_maps_multigpu.py:193-216; eight launch threads of ~500 launches per
    step each should neither migrate nor share cores).  Falls back to an even split of the allowed CPUs when sysfs says nothing.
    `nodes` (test hook): NUMA node per local rank; `allowed`: the CPU list to carve from.  Returns what was done (for the rank report)."""
    allowed = sorted(os.sched_getaffinity(0)) if allowed is None else sorted(allowed)
    if nodes is None:
        nodes = [gpu_numa_node(i) for i in range(local_world)]
    how, mine = 'even split of the allowed CPUs', None
    node = nodes[local] if local < len(nodes) else None
    if node is not None:
        try:
            node_cpus = [c for c in parse_cpulist(open('/sys/devices/system/node/node%d/cpulist' % node).read()) if c in set(allowed)]
            peers = [i for i in range(local_world) if nodes[i] == node]
            if len(node_cpus) >= len(peers):
                mine = cpu_slices(node_cpus, len(peers))[peers.index(local)]
                how = 'NUMA node %d split among %d rank(s)' % (node, len(peers))
        except Exception:
            mine = None
    if mine is None:
        try:
            mine = cpu_slices(allowed, local_world)[local]
        except ValueError as ex:        # fewer allowed CPUs than local ranks: leave the affinity alone and say so
            return {'cpus': None, 'n_cpus': len(allowed), 'numa_node': node, 'how': 'not bound (%s)' % (ex,)}
    try:
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(len(mine), torch.get_num_threads())))
    except Exception as ex:
        how += ' (sched_setaffinity failed: %r)' % (ex,)
    return {'cpus': '%d-%d' % (mine[0], mine[-1]) if mine == list(range(mine[0], mine[-1] + 1)) else ','.join(map(str, mine)),
            'n_cpus': len(mine), 'numa_node': node, 'how': how}


def gather_rank_rates(rate, device=None):
    """Every rank's own rate (units/s over ITS time for the timed steps) -> {'per_rank', 'min', 'max', 'spread'} on every rank;
    spread = (max - min) / max: a straggler that the whole-job figure (max-over-ranks time) hides shows up here."""
    rates = [float(rate)]
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([float(rate)], dtype=torch.float64, device=device if dist.get_backend() == 'nccl' else 'cpu')
        outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(outs, t)
        rates = [float(o.item()) for o in outs]
    mx, mn = max(rates), min(rates)
    return {'per_rank': rates, 'min': mn, 'max': mx, 'spread': (mx - mn) / mx if mx > 0 else 0.0}
