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


class Comm(object):
    """What bench.py --gpus N needs between its ranks, made to survive a collective backend that does not come up.

    Two channels.  (1) A TCPStore of its own (MASTER_PORT + 17, rank 0 serves), created BEFORE torch.distributed is initialised and used for
    everything that is small: per-rank reports, exception texts, agreement flags, and -- when the collective backend is unusable -- barriers and
    the max-over-ranks time.  It does not depend on RCCL.  (2) torch.distributed (backend "nccl" = RCCL over xGMI; gloo in the tests) for the one
    payload that matters, the packed parameter arena, and for the barriers around the timed region while it is healthy.  Rendezvous and
    every wait are bounded (XFR_DIST_TIMEOUT seconds, default 180): a rank that never shows up becomes an error text on rank 0, not a hang.
    The reference's pool does the same per job (eval/generate_inpaintinggame_wb_saliency_maps_multigpu.py:101-118: try / except around every
    job, failed jobs listed at the end, :193-224)."""

    def __init__(self, backend=None, timeout_s=None):
        import datetime
        self.rank, self.world, self.local = dist_env()
        self.timeout_s = float(timeout_s if timeout_s is not None else os.environ.get('XFR_DIST_TIMEOUT', '180'))
        self.op_timeout_s = 0.4 * self.timeout_s
        self.store = None
        self.collective_ok = False
        self.init_error = None
        self.notes = []                  # non-fatal conditions worth printing (a failed broadcast that the fallback absorbed)
        self.weights_via = 'local_pack' if self.world == 1 else None
        self._seq = 0
        if self.world == 1:
            return
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        td = datetime.timedelta(seconds=self.timeout_s)
        # a collective may use up 40 % of the budget before it gives up; the store waits (reports, votes, barriers) get all of it, so that a rank which
        # arrives at a vote only after its collective has timed out is still waited for
        self.op_timeout_s = 0.4 * self.timeout_s
        td_op = datetime.timedelta(seconds=self.op_timeout_s)
        self.store = dist.TCPStore(os.environ['MASTER_ADDR'], int(os.environ['MASTER_PORT']) + 17, self.world, self.rank == 0, timeout=td)
        try:
            if os.environ.get('XFR_TEST_FAIL_INIT') == '1':
                raise RuntimeError('XFR_TEST_FAIL_INIT: simulated rendezvous failure')
            if backend is None:
                backend = os.environ.get('XFR_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
            if backend == 'nccl':
                torch.cuda.set_device(self.local)
            if not dist.is_initialized():
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, timeout=td_op)
            self.collective_ok = True
        except Exception as ex:        # noqa: BLE001 -- whatever the backend throws: the ranks go on over the store
            self.init_error = repr(ex)
        # one decision for all ranks: a collective backend is only used if EVERY rank has it
        self.collective_ok = self.agree(self.collective_ok)

    # -- the store channel ---------------------------------------------------------------------------------------------------
    def gather_objects(self, obj, tag, fail_fast=True, wait_s=None):
        """[obj of rank 0, ..., obj of rank world-1] on every rank (JSON through the store); a missing rank yields {'missing_rank': r}."""
        import json
        if self.world == 1:
            return [obj]
        import time
        self.store.set('%s/%d' % (tag, self.rank), json.dumps(obj))
        out = []
        for r in range(self.world):
            key, t0, val = '%s/%d' % (tag, r), time.time(), None
            while val is None:
                # (store.check polls without blocking and without the warning c10d prints for every timed-out wait: a waiting rank must stay quiet)
                if self.store.check([key]):
                    val = json.loads(self.store.get(key).decode())
                    break
                if fail_fast and r != self.rank and self.store.check(['error/%d' % r]):
                    raise RuntimeError('rank(s) [%d] failed: %s' % (r, self.store.get('error/%d' % r).decode().strip().splitlines()[-1]))
                if time.time() - t0 > (wait_s if wait_s is not None else self.timeout_s):
                    val = {'missing_rank': r, 'error': 'no %s from rank %d within %.0f s' % (tag, r, wait_s if wait_s is not None else self.timeout_s)}
                    break
                time.sleep(0.002)
            out.append(val)
        return out

    def next_tag(self, name):
        """A store tag no earlier gather of this communicator has used (every rank calls in the same order): keys are never deleted, so a second
        gather under a fixed tag would read the first one's values at once."""
        self._seq += 1
        return '%s%d' % (name, self._seq)

    def agree(self, flag):
        """True iff every rank passed True (a rank that never answers counts as False)."""
        if self.world == 1:
            return bool(flag)
        self._seq += 1
        votes = self.gather_objects(bool(flag), 'agree%d' % self._seq)
        return all(v is True for v in votes)

    def barrier(self):
        """A store barrier, followed by torch.distributed's while the collective backend is healthy."""
        if self.world == 1:
            return
        # the store barrier comes first either way: it can see a rank that has reported an exception (and then raises here instead of waiting
        # for a collective that rank will never join)
        self._seq += 1
        key = 'barrier%d' % self._seq
        self.store.add(key, 1)
        import time
        t0 = tc = time.time()
        while int(self.store.add(key, 0)) < self.world:
            now = time.time()
            if now - tc > 0.05:
                tc = now
                errs = self.collect_errors()
                errs.pop(self.rank, None)
                if errs:
                    raise RuntimeError('rank(s) %s failed: %s' % (sorted(errs), '; '.join(e.strip().splitlines()[-1] for e in errs.values())))
            if now - t0 > self.timeout_s:
                raise RuntimeError('barrier timed out after %.0f s' % self.timeout_s)
            time.sleep(0.002)          # (2 ms: the timed region that follows starts after a device synchronise on every rank anyway)
        if self.collective_ok:
            dist.barrier()

    def max_float(self, x, device=None):
        if self.world == 1:
            return float(x)
        if self.collective_ok:
            t = torch.tensor([float(x)], dtype=torch.float64, device=device if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        self._seq += 1
        return max(float(v) for v in self.gather_objects(float(x), 'max%d' % self._seq) if not isinstance(v, dict))

    def report_error(self, text):
        if self.store is not None:
            try:
                self.store.set('error/%d' % self.rank, text)
            except Exception:          # noqa: BLE001
                pass

    def collect_errors(self, wait_s=0.0):
        """{rank: exception text} of the ranks that reported one (rank 0 calls this before it prints its line)."""
        import time
        errs = {}
        if self.store is None:
            return errs
        t0 = time.time()
        while True:
            for r in range(self.world):
                try:
                    if r not in errs and self.store.check(['error/%d' % r]):
                        errs[r] = self.store.get('error/%d' % r).decode()
                except Exception:      # noqa: BLE001
                    pass
            if time.time() - t0 >= wait_s:
                return errs
            time.sleep(0.05)

    # -- the payload ---------------------------------------------------------------------------------------------------------
    def load_weights(self, engine, state_dict_fn, src=0):
        """Rank `src` packs, the arena travels with ONE broadcast; if the collective backend is down or the broadcast fails on any rank, EVERY rank
        packs from the seed locally (the pack is deterministic: the arena checksums of the per-rank report must still agree).  Sets
        self.weights_via: 'local_pack' (one rank), 'broadcast', or 'local_pack_fallback'."""
        if self.world == 1:
            engine.load_weights(state_dict_fn())
            return engine
        ok = self.collective_ok
        err = None
        if ok:
            try:
                if os.environ.get('XFR_TEST_FAIL_BROADCAST') == '1' and self.rank != src:
                    raise RuntimeError('XFR_TEST_FAIL_BROADCAST: simulated broadcast failure')
                if self.rank == src:
                    engine.load_weights(state_dict_fn())
                arena = engine.weight_arena()
                work = dist.broadcast(arena, src=src, async_op=True)
                import datetime
                work.wait(datetime.timedelta(seconds=self.op_timeout_s))
                if arena.is_cuda:
                    torch.cuda.synchronize()
            except Exception as ex:    # noqa: BLE001
                ok, err = False, repr(ex)
        if self.agree(ok):
            if self.rank != src:
                engine.mark_weights_loaded()
            self.weights_via = 'broadcast'
            return engine
        if err:
            sys_note = 'weight broadcast failed on rank %d: %s' % (self.rank, err)
            self.notes.append(sys_note)
        self.collective_ok = False          # a collective that failed half way leaves the backend in an unknown state: the store from here on
        engine.load_weights(state_dict_fn())
        self.weights_via = 'local_pack_fallback'
        return engine

    def close(self):
        if self.world > 1 and self.store is not None:
            # Check out through the store.  It lives in rank 0's process: a rank that is still leaving its last store barrier (it polls every 2 ms)
            # finds the connection closed if rank 0 has already exited -- without a healthy collective backend nothing else orders the exits.  Rank 0
            # waits for every rank that has not reported an exception, for at most 30 s.
            try:
                import time
                self.store.add('checkout', 1)
                if self.rank == 0:
                    t0, tc, gone = time.time(), 0.0, 0
                    while time.time() - t0 < min(self.timeout_s, 30.0):
                        if time.time() - tc > 0.25:       # ranks that reported an exception never check out
                            tc = time.time()
                            gone = len([r for r in self.collect_errors() if r != 0])
                        if int(self.store.add('checkout', 0)) >= self.world - gone:
                            break
                        time.sleep(0.002)
            except Exception:          # noqa: BLE001 -- leaving anyway
                pass
        if self.world > 1 and dist.is_initialized():
            try:
                if self.collective_ok:
                    dist.barrier()
                dist.destroy_process_group()
            except Exception:          # noqa: BLE001
                pass


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


def gather_rank_rates(rate, device=None, comm=None):
    """Every rank's own rate (units/s over ITS time for the timed steps) -> {'per_rank', 'min', 'max', 'spread'} on every rank;
    spread = (max - min) / max: a straggler that the whole-job figure (max-over-ranks time) hides shows up here.  `comm` (a Comm): through its
    store, so that the figure exists even when the collective backend does not."""
    rates = [float(rate)]
    if comm is not None and comm.world > 1:
        rates = [float(v) if not isinstance(v, dict) else float('nan') for v in comm.gather_objects(float(rate), comm.next_tag('rank_rate'))]
    elif dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([float(rate)], dtype=torch.float64, device=device if dist.get_backend() == 'nccl' else 'cpu')
        outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(outs, t)
        rates = [float(o.item()) for o in outs]
    mx, mn = max(rates), min(rates)
    return {'per_rank': rates, 'min': mn, 'max': mx, 'spread': (mx - mn) / mx if mx > 0 else 0.0}
