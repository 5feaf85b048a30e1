#!/usr/bin/env python
"""bench.py -- triplet-contrastive-EBP saliency maps/sec, ResNet-101 224x224 (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of B synthetic triplets per GPU, inputs resident in HBM:
    encode(mates[B]); encode(nonmates[B]);                      (demo/test_whitebox.py:71-72)
    classifier rows = encoding / 2500                           (demo/test_whitebox.py:129)
    contrastive_ebp(probes[B], 0, 1) -> B saliency maps 112x112 (whitebox.py:506-527), per-triplet classifier
Workload = BASELINE.json configs[1]: ResNet-101 triplet contrastive EBP, batch=32 synthetic 224x224 triplets per
GPU, ebp_subtree_mode 'affineonly_with_prior' (the demo default), fp32 throughout.  `--model resnet50_128 | lightcnn` run
configs[2] / configs[3] under the same contract and emit the same objects.

Multi-GPU: one process per GPU (torch.distributed.run), triplets sharded embarrassingly, weights packed on rank 0
and broadcast once over RCCL; no steady-state collective => "scaling": "weak" (B triplets per GPU).

At N = 1 the default run also times BASELINE.json configs[2] (ResNet-50-128d truncated contrastive, B=64, 'norelu') and configs[3]
(Light-CNN-29v2 EBP, B=128, 'affineonly') for a few steps each and appends them as `"secondary": [...]` (--no-secondary skips them;
absent at N > 1 and when --model selects one of them as the main workload).

Prints ONE JSON line on rank 0.  The `roofline` object describes the TIMED schedule: `achieved` is the algorithmic FLOPs over
the union of the GEMM launches' busy intervals, taken from timestamps the kernels themselves record while the step runs on its
three streams exactly as it was timed (xfr_amd/tuning.py); the one-stream figure rocprofv3 can reproduce is `frac_serial`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_FWD = {'resnet101': 14.419e9, 'resnet50_128': 7.712e9, 'lightcnn': 7.275e9}   # 2*MAC over conv+linear
PEAK_BF16_MFMA = 2516.6e12        # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, 1024 FLOP/clk/SIMD * 1024 SIMD * 2.4 GHz (dense)
PEAK_F32_MFMA = 157.3e12         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD * 1024 SIMD * 2.4 GHz
ROW0_COS = 0.99999               # sample 0 against the map the reference computes for the same unit of work (tests/golden/golden_bench.npz, golden_synth.npz): the parity tests' cosine bar


def fixture_cosine(sal0, key):
    """Cosine between the engine's map for sample 0 of rank 0's batch and the committed reference map of that triplet."""
    import numpy as np
    want = None
    for name in ('golden_bench.npz', 'golden_synth.npz'):
        f = os.path.join(ROOT, 'tests', 'golden', name)
        if os.path.exists(f):
            z = np.load(f)
            if key + '/map' in z.files:
                want = z[key + '/map'].astype(np.float64).ravel()
    if want is None:
        return None
    got = sal0.detach().cpu().numpy().astype(np.float64).ravel()
    return float(got @ want / max(np.linalg.norm(got) * np.linalg.norm(want), 1e-300))


def port_vs_reference(root):
    """How the port relates to the real reference on identical hardware.  The reference cannot
    travel to the GPU box, so the ratio is measured in the build container (tools/measure_port_vs_reference.py: same
    triplets, same thread count, interleaved) and committed under profiles/rNN/."""
    import glob
    for d in reversed(sorted(glob.glob(os.path.join(root, 'profiles', 'r[0-9]*')))):
        f = os.path.join(d, 'port_vs_reference.json')
        if os.path.exists(f):
            j = json.load(open(f))
            return {'port_vs_reference': j['port_vs_reference'],
                    'port_vs_reference_source': '%s: reference %.3f maps/s, port %.3f maps/s, %d threads, build container'
                                                % (os.path.relpath(f, root), j['reference_maps_s'], j['port_maps_s'], j['threads'])}
    return {}


def time_port(unit, what, n_units, budget_s):
    """The oracle (kind "port": hook-free CPU restatement of the reference, bit-exact against it in the build container) timed on
    this box's host cores on a bounded sample of the same workload: whole units until ~budget_s of CPU time is spent (at least
    one).  The reference's path is batch 1 (whitebox.py:512); its convolutions stop scaling long before a 128-core host is used up
    (measured on the GPU box: 8.7 s per ResNet-101 triplet with 128 threads, 1.0 s with 32, 0.46 s with 16), so the thread count
    is chosen by a short probe and reported as `cores`."""
    import torch
    ncpu = os.cpu_count() or 1
    usable = host_cpu_budget()['usable']
    best_t, best_dt = None, None
    for th in sorted({max(1, min(ncpu, c)) for c in (8, 16, 32, usable)}):
        torch.set_num_threads(th)
        t0 = time.time()
        unit(0)
        d = time.time() - t0
        if best_dt is None or d < best_dt:
            best_t, best_dt = th, d
    torch.set_num_threads(best_t)
    n = 0
    t0 = time.time()
    while True:
        unit(n % n_units)
        n += 1
        if time.time() - t0 > budget_s or n >= 64:
            break
    dt = time.time() - t0
    return {'value': n / dt, 'unit': 'maps/s', 'cores': int(best_t), 'kind': 'port',
            'sample': '%d %s, batch 1, %.1f s, %d threads (best of 8/16/32/%d; host shows %d logical CPUs, grants %d)' % (n, what, dt, best_t, usable, ncpu, usable)}


def cpu_worker_main(model, seconds, threads, batch, mode):
    """`python bench.py --cpu-worker ...`: one worker of the whole-host CPU baseline.  Builds the port for `model`, runs units until
    `seconds` have passed after a start line read from stdin (so that all workers measure the same window), prints the count."""
    import torch
    torch.set_num_threads(threads)

    class A(object):
        pass
    a = A()
    a.model, a.batch, a.mode = model, batch, mode
    W = make_workload(a, torch.device('cpu'), 0, cpu_only=True)
    unit, n_units = W.cpu_unit()
    unit(0)                                   # warm: thread pool, allocator, oneDNN primitives
    sys.stdout.write('ready\n'); sys.stdout.flush()
    sys.stdin.readline()
    n, t0 = 0, time.time()
    while time.time() - t0 < seconds:
        unit(n % n_units)
        n += 1
    print(json.dumps({'units': n, 'seconds': time.time() - t0}))


def host_cpu_budget():
    """CPUs this process may really use: the smaller of the affinity mask and the cgroup CPU quota (cpu.max / cfs_quota_us).  The GPU
    boxes of this pool show 256 hardware threads to os.cpu_count() under a quota of 16 CPUs: threads beyond the quota only get throttled
    (round 4: 16 processes x 16 threads finished 16 units in 28.7 s where ONE such process finishes 49 in 20 s)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    return {'logical': os.cpu_count() or 1, 'affinity': n, 'cgroup_quota': quota, 'usable': int(max(1, min(n, quota if quota else n)))}


def whole_host(model, batch, mode, threads, budget_s):
    """north_star: "next to the reference's CPU-only path timed on the same box's host cores".  One worker cannot use the host (the
    batch-1 convolutions stop scaling at ~16 threads), so P = cpu_count // threads worker PROCESSES of `threads` threads each run the
    port side by side for `budget_s` seconds; value = all units finished / the common window."""
    import subprocess
    budget = host_cpu_budget()
    ncpu = budget['usable']
    P = max(1, ncpu // max(threads, 1))
    if P == 1:
        return {'processes': 1, 'host_cpus': budget,
                'note': 'the host grants this process %d CPUs (cgroup quota / affinity; %d logical): the %d-thread figure above IS the whole-host figure' % (
                    ncpu, budget['logical'], threads)}
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-worker', '--model', model, '--cpu-worker-seconds', str(budget_s),
           '--cpu-worker-threads', str(threads)] + (['--batch', str(batch)] if batch else []) + (['--mode', mode] if mode else [])
    procs = [subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for _ in range(P)]
    try:
        for q in procs:                        # every worker has built its model and run one unit
            if q.stdout.readline().strip() != 'ready':
                raise RuntimeError('a CPU worker died during start-up')
        t0 = time.time()
        for q in procs:
            q.stdin.write('go\n'); q.stdin.flush()
        outs = [json.loads(q.stdout.readline()) for q in procs]
        dt = time.time() - t0
    except Exception as ex:
        for q in procs:
            q.kill()
        return {'error': repr(ex)}
    finally:
        for q in procs:
            try:
                q.wait(timeout=30)
            except Exception:
                q.kill()
    units = sum(o['units'] for o in outs)
    return {'value': units / dt, 'unit': 'maps/s', 'cores': P * threads, 'processes': P, 'threads_per_process': threads, 'kind': 'port', 'host_cpus': budget,
            'sample': '%d units by %d processes x %d threads in %.1f s (host has %d hardware threads)' % (units, P, threads, dt, ncpu)}


def pmc_traffic(root, tag):
    """HBM bytes per GEMM launch from the committed PMC passes of this same command (profiles/rNN/pmc_FETCH_SIZE<tag>.txt,
    pmc_WRITE_SIZE<tag>.txt: separate rocprofv3 --pmc runs, KB summed over the dispatches).  FETCH_SIZE is doubled: on gfx950 it
    reports half the bytes of 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM section); the dword-wide im2col reads of the KxK
    layers are not calibrated, so this is an upper bound.  Counters cannot be read from inside the timed process, hence the file;
    {} if no profile has been committed."""
    import glob
    import re
    for d in reversed(sorted(glob.glob(os.path.join(root, 'profiles', 'r[0-9]*')))):
        try:
            vals = {}
            for name in ('FETCH_SIZE', 'WRITE_SIZE'):
                n, v = 0, 0.0
                for ln in open(os.path.join(d, 'pmc_%s%s.txt' % (name, tag))):
                    m = re.match(r'conv_gemm(?:_ks|_split)?_kernel\s+dispatches\s+(\d+)\s+.*%s=([0-9.e+]+)' % name, ln)
                    if m:
                        n += int(m.group(1))
                        v += float(m.group(2))
                vals[name] = (n, v)
            (n, f), (n2, w) = vals['FETCH_SIZE'], vals['WRITE_SIZE']
            if n == 0 or n2 == 0:
                continue
            return {'traffic': (2.0 * f * 1024 / n + w * 1024 / n2), 'traffic_unit': 'HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)',
                    'traffic_source': '%s/pmc_FETCH_SIZE%s.txt, pmc_WRITE_SIZE%s.txt' % (os.path.relpath(d, root), tag, tag)}
        except (OSError, KeyError, ValueError):
            continue
    return {}


def pmc_mfma(root, tag, launches_per_step, ms_step, clk_ghz):
    """MFMA-pipe utilisation of the GEMM kernels from the committed PMC pass of this same command (profiles/rNN/pmc_mfma<tag>.txt, one-stream schedule under
    rocprofv3): sum SQ_VALU_MFMA_BUSY_CYCLES / (sum GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), all GEMM kernels and per kernel -- and what the same busy
    cycles per step are of the TIMED step (busy cycles do not depend on the schedule; the step's cycles = its duration x the shader clock it held)."""
    import glob
    import re
    for d in reversed(sorted(glob.glob(os.path.join(root, 'profiles', 'r[0-9]*')))):
        try:
            rows = {}
            for ln in open(os.path.join(d, 'pmc_mfma%s.txt' % tag)):
                m = re.match(r'(conv_gemm(?:_ks|_split)?_kernel)\s+dispatches\s+(\d+)\s+(.*)', ln)
                if m:
                    kv = dict(t.split('=') for t in m.group(3).split())
                    rows[m.group(1)] = (int(m.group(2)), float(kv['SQ_VALU_MFMA_BUSY_CYCLES']), float(kv['GRBM_GUI_ACTIVE']))
            if not rows:
                continue
            n = sum(r[0] for r in rows.values())
            busy = sum(r[1] for r in rows.values())
            act = sum(r[2] for r in rows.values())
            out = {'mfma_util_serial_pmc': busy / (act / 8.0 * 1024.0),
                   'mfma_util_serial_pmc_by_kernel': {k: r[1] / (r[2] / 8.0 * 1024.0) for k, r in sorted(rows.items())},
                   'mfma_util_source': '%s/pmc_mfma%s.txt (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024), one-stream schedule under rocprofv3)' % (os.path.relpath(d, root), tag)}
            if launches_per_step and ms_step and clk_ghz:
                steps = n / float(launches_per_step)
                out['mfma_util_timed_estimate'] = (busy / steps) / (ms_step * 1e-3 * clk_ghz * 1e9 * 1024.0)
                out['mfma_util_timed_how'] = ('the PMC pass\'s MFMA busy cycles per step (%d dispatches / %.0f launches per step = %.1f steps) over the timed step\'s '
                                              'cycles (%.3f ms x %.2f GHz x 1024 SIMDs)' % (n, launches_per_step, steps, ms_step, clk_ghz))
            return out
        except (OSError, KeyError, ValueError):
            continue
    return {}


def one_triplet_latency(W, reps=30):
    """The reference's API is one image per call (Whitebox.contrastive_ebp(img), whitebox.py:506-527; demo/test_whitebox.py:124-133 brings one triplet at a
    time): wall time of ONE triplet through the same entry point -- two encodes and the contrastive sweep, batch 1, the device synchronised after every
    call, fresh (not resident-declared) inputs, every library default.  Median and p90 over `reps` calls after 5 warm-up calls.  (Replaying the
    call from a hipGraph recording was measured in round 6 and is not offered: profiles/r6/experiments/recorded_runs.txt.)"""
    import torch
    if getattr(W, 'one', None) is None:
        return None

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ts, th = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            s = fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
            th.append(1e3 * (t1 - t0))       # the calling thread's share: enqueueing the call
        ts.sort()
        th.sort()
        return {'ms_median': ts[len(ts) // 2], 'ms_p90': ts[int(0.9 * (len(ts) - 1))], 'host_ms_median': th[len(th) // 2]}, s
    out, s = timed(W.one)
    out.update({'calls': reps, 'outputs_ok': bool(torch.isfinite(s).all().item()) and abs(float(s.sum().item()) - 1.0) < 1e-3,
                'what': ('one image per call (EBP over the 80013-way classifier)' if W.model == 'lightcnn' else 'one triplet per call (2 encodes + contrastive EBP)') +
                   ", device synchronised after each call, library defaults; host_ms = the calling thread's time inside the call"})
    return out


def chain_stats():
    import ctypes
    from xfr_amd import _lib
    c, i, n = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32()
    _lib.check(_lib.load().xfr_chain_epilogue_stats(ctypes.byref(c), ctypes.byref(i), ctypes.byref(n)))
    return c.value, i.value, n.value


class Workload(object):
    """One BASELINE.json configuration: engine, resident inputs, the step, its checks and its CPU baseline."""
    pass


def make_u8_step(eng, imgs, mean, B, enc_t, pct, dev):
    """The same triplet step fed the way a caller holding decoded crops would feed it (xfr_triplet_contrastive_u8_host): the 3B images as uint8 H x W x 3
    in PINNED host memory, two buffers used alternately (a real caller decodes the next batch into the one the engine is done copying), the
    host-to-device copy and the on-device preprocessing inside the step -- on the engine's own copy stream and staging buffers, so that they overlap
    the previous step's sweep without any residency promise.  imgs: the bench's fp32 images (mean subtracted); their uint8 versions are round(img + mean)."""
    import torch
    m = torch.tensor(mean, dtype=torch.float32).reshape(1, 3, 1, 1)
    u8 = (imgs + m).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    bufs = [(u8[2 * B:].clone().pin_memory(), u8[:2 * B].clone().pin_memory()) for _ in range(2)]
    eng.set_u8_preprocess('sub_mean', 3, tuple(float(v) for v in mean), None)
    state = {'i': 0}

    def step():
        p, g = bufs[state['i'] & 1]
        state['i'] += 1
        if state['i'] > 2:
            eng.wait_inputs_copied()          # the buffer about to be "refilled" (used two calls ago) has been copied: copies run in order
        return eng.triplet_contrastive_u8_host(p, g, enc_t, 1.0 / 2500.0, pct)
    return step


def make_workload(args, dev, rank, cpu_only=False, comm=None):
    """cpu_only: no engine, no device -- just the inputs and the CPU port's unit of work (the whole-host baseline's workers)."""
    import torch
    from xfr_amd import shard, synth
    W = Workload()
    W.model = args.model
    sd_holder = {}
    if not cpu_only:
        from xfr_amd.engine import Engine
    if args.model == 'resnet101':
        from xfr_amd.models import resnet
        W.mode = args.mode or 'affineonly_with_prior'
        B = W.B = args.batch or 32
        bb = resnet.ResNet([3, 4, 23, 3], num_classes=2)   # fc2 is replaced by the per-triplet classifier anyway
        make_sd = lambda: sd_holder.setdefault('sd', synth.synth_state_dict(bb, seed=0, recipe='mild'))   # noqa: E731
        imgs = synth.bench_images(B, (3, 224, 224), seed=1234 + rank, mean=resnet.MEAN_RGB)
        if not cpu_only:
            prog = bb.build_program()
            eng = Engine(prog, 2 * B, dev)                 # the two encode batches of a step run as one 2B-image forward
            mates, nonmates, probes = imgs[0:B].to(dev), imgs[B:2 * B].to(dev), imgs[2 * B:3 * B].to(dev)
            gallery = torch.cat((mates, nonmates), dim=0)  # [2B,3,224,224] resident in HBM
            enc_t = prog.marks['encode']
            W.pipeline = 1
            W.step = lambda ready=True: eng.triplet_contrastive(probes, gallery, enc_t, 1.0 / 2500.0, None, inputs_ready=ready)   # noqa: E731
            W.u8_step = make_u8_step(eng, imgs, resnet.MEAN_RGB, B, enc_t, None, dev)
            g1, p1 = torch.stack((gallery[0], gallery[B])).contiguous(), probes[:1].contiguous()
            W.one = lambda: eng.triplet_contrastive(p1, g1, enc_t, 1.0 / 2500.0, None)   # noqa: E731

            def step_inputs(n):        # n triplets per call, fresh (not resident-declared) inputs: tools/one_triplet_probe.py --triplets
                gn, pn = torch.cat((gallery[:n], gallery[B:B + n])).contiguous(), probes[:n].contiguous()
                return lambda: eng.triplet_contrastive(pn, gn, enc_t, 1.0 / 2500.0, None)
            W.step_inputs = step_inputs
        W.flop_per_unit = 6 * F_FWD['resnet101']           # 2 encodes + true fwd + relu(W) fwd + 2 backward-data sweeps = 86.51 GFLOP
        W.metric = 'triplet-contrastive-EBP saliency maps/sec, ResNet-101 224x224'
        W.work = ('ResNet-101 triplet contrastive EBP, batch=%d synthetic 224x224 triplets per GPU (2 encodes + contrastive_ebp per '
                  'triplet), mode %s, eps 1e-16' % (B, W.mode))
        W.fixture = 'bench/r101' if (B == 32 and W.mode == 'affineonly_with_prior') else None
        W.pmc_tag = '' if (B == 32 and W.mode == 'affineonly_with_prior') else None
        W.cpu_what = 'ResNet-101 triplet(s) (2 encodes + contrastive_ebp each)'

        def cpu_unit():
            from oracle import ebp_oracle as O
            ow = O.OracleWhitebox('stresnet101', make_sd(), ('hooked', None), W.mode)
            pm, pn, pp = imgs[0:B], imgs[B:2 * B], imgs[2 * B:3 * B]

            def unit(i):
                ow.set_triplet_classifier(ow.encode(pm[i:i + 1]) / 2500.0, ow.encode(pn[i:i + 1]) / 2500.0)
                ow.contrastive_ebp(pp[i:i + 1], 0, 1)
            return unit, B
    elif args.model == 'resnet50_128':
        from xfr_amd.models import resnet50_128
        W.mode = args.mode or 'norelu'
        B = W.B = args.batch or 64
        bb = resnet50_128.Resnet50_128()
        make_sd = lambda: sd_holder.setdefault('sd', synth.synth_state_dict(bb, seed=0))   # noqa: E731
        imgs = synth.bench_images(B, (3, 224, 224), seed=1234 + rank, mean=(131.0912, 103.8827, 91.4953))
        if not cpu_only:
            prog = bb.build_program()
            eng = Engine(prog, 2 * B, dev)
            gallery, probes = imgs[:2 * B].to(dev).contiguous(), imgs[2 * B:].to(dev).contiguous()
            enc_t = prog.marks['encode']
            W.pipeline = 1
            W.step = lambda ready=True: eng.triplet_contrastive(probes, gallery, enc_t, 1.0 / 2500.0, 20.0, inputs_ready=ready)   # noqa: E731
            W.u8_step = make_u8_step(eng, imgs, (131.0912, 103.8827, 91.4953), B, enc_t, 20.0, dev)
            g1, p1 = torch.stack((gallery[0], gallery[B])).contiguous(), probes[:1].contiguous()
            W.one = lambda: eng.triplet_contrastive(p1, g1, enc_t, 1.0 / 2500.0, 20.0)   # noqa: E731
        W.flop_per_unit = 6 * F_FWD['resnet50_128']
        W.metric = 'triplet truncated-contrastive-EBP (20 %) saliency maps/sec, VGGFace2 ResNet-50-128d 224x224'
        W.work = 'ResNet-50-128d truncated contrastive EBP, batch=%d synthetic triplets per GPU, mode %s' % (B, W.mode)
        W.fixture = 'bench/r50' if (B == 64 and W.mode == 'norelu') else None
        W.pmc_tag = '_r50' if (B == 64 and W.mode == 'norelu') else None
        W.cpu_what = 'ResNet-50-128d triplet(s) (2 encodes + truncated_contrastive_ebp each)'

        def cpu_unit():
            from oracle import ebp_oracle as O
            ow = O.OracleWhitebox('resnet50_128', make_sd(), ('hooked', None), W.mode)
            g, pp = imgs[:2 * B], imgs[2 * B:]

            def unit(i):
                ow.set_triplet_classifier(ow.encode(g[i:i + 1]) / 2500.0, ow.encode(g[B + i:B + i + 1]) / 2500.0)
                ow.truncated_contrastive_ebp(pp[i:i + 1], 0, 1, percentile=20)
            return unit, B
    else:
        from xfr_amd.models import lightcnn
        W.mode = args.mode or 'affineonly'
        B = W.B = args.batch or 128
        bb = lightcnn.LightCNN_29Layers_v2(num_classes=80013)
        make_sd = lambda: sd_holder.setdefault('sd', synth.synth_state_dict(bb, seed=0))   # noqa: E731
        xs = synth.synth_images(B, (1, 128, 128), seed=1234 + rank, scale255=False)
        if not cpu_only:
            prog = bb.build_program()
            eng = Engine(prog, B, dev)
            x = xs.to(dev)
            seed = torch.zeros((1, B, 80013), device=dev)
            seed[0, :, 0] = 1.0
            cls_t = prog.marks['classify']
            W.pipeline = 6          # forward of step i+1 under the backward of step i; x is resident (inputs_ready below).  Bit 2: THREE forward
                                    # slots -- this schedule has one forward stream, which then runs up to two steps ahead: +1.8 % (the ResNets'
                                    # three-stream step is level with it: profiles/r4/experiments/pipeline_slots_ab.txt)

            def step(ready=True):
                _, pooled = eng.ebp(x, cls_t, seed, want_mwp=False, want_pooled=True, inputs_ready=ready)
                return eng.mwp_to_saliency(pooled[0])
            W.step = step
            x1, seed1 = x[:1].contiguous(), seed[:, :1].contiguous()

            def one():             # one image per call, like Whitebox.ebp(img, P) (whitebox.py:490-498)
                _, pooled = eng.ebp(x1, cls_t, seed1, want_mwp=False, want_pooled=True)
                return eng.mwp_to_saliency(pooled[0])
            W.one = one
        # the engine only runs the relu(W) forward where a hook divides by X ('affineonly' needs none): 2 F_fwd executed, 3 F_fwd in
        # the modes that divide by a Split's X -- the roofline uses the executed GEMM FLOPs, capped by SURVEY's 3 F_fwd
        W.flop_per_unit = 3 * F_FWD['lightcnn']
        W.metric = 'EBP saliency maps/sec, Light-CNN-29v2 128x128 (80013-way hooked classifier)'
        W.work = 'Light-CNN-29v2 excitation backprop, batch=%d synthetic images per GPU, mode %s' % (B, W.mode)
        W.fixture = 'bench/lcnn' if (B == 128 and W.mode == 'affineonly') else None     # image 0 through the real reference (tests/golden/make_golden_synth.py)
        W.pmc_tag = '_lcnn' if (B == 128 and W.mode == 'affineonly') else None
        W.cpu_what = 'Light-CNN-29v2 ebp call(s) over the 80013-way classifier'

        def cpu_unit():
            from oracle import ebp_oracle as O
            ow = O.OracleWhitebox('lightcnn29v2', make_sd(), ('hooked', None), W.mode)
            P = torch.zeros((1, 80013))
            P[0, 0] = 1.0
            return (lambda i: ow.ebp(xs[i:i + 1], P)), B
    W.cpu_unit = cpu_unit
    W.batch_arg = args.batch

    def cpu(budget, with_whole_host=True):
        unit, n_units = cpu_unit()
        out = time_port(unit, W.cpu_what, n_units, budget)
        if W.model == 'resnet101':
            out.update(port_vs_reference(ROOT))
        if with_whole_host:
            out['whole_host'] = whole_host(W.model, W.batch_arg, W.mode, out['cores'], budget)
        return out
    W.cpu_baseline = cpu
    W.sd_holder = sd_holder
    if cpu_only:
        return W
    W.eng = eng
    if comm is not None:
        comm.load_weights(eng, make_sd, src=0)         # rank 0 packs, everybody receives the arena over RCCL (or, if that fails anywhere, packs locally)
    else:
        shard.load_and_broadcast(eng, make_sd, src=0)
    eng.set_mode(W.mode)
    return W


def rank_report(eng, comm, binding=None):
    """What proves N ranks and one broadcast in the driver's log: per rank the device, the checksum of the packed parameter arena it
    ended up with (equal on all ranks), how it got there (broadcast | local_pack_fallback) and the RCCL version; gathered on every rank
    (through the store: no collective needed), echoed on stderr by every rank."""
    import torch
    import torch.distributed as dist
    rank, local, world = comm.rank, comm.local, comm.world
    arena = eng.weight_arena()
    n8 = (arena.numel() // 8) * 8
    crc = int(arena[:n8].view(torch.int64).sum().item()) & 0xffffffffffff
    try:
        rccl = '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        rccl = None
    from xfr_amd import shard
    me = {'rank': rank, 'device': local, 'visible_device': shard.normalize_gpus([local])[0], 'device_name': torch.cuda.get_device_name(local),
          'arena_bytes': int(arena.numel()), 'arena_checksum48': '%012x' % crc, 'rccl': rccl,
          'backend': dist.get_backend() if dist.is_initialized() else None, 'weights_via': comm.weights_via}
    if comm.init_error:
        me['collective_init_error'] = comm.init_error
    if binding is not None:
        me['cpu_binding'] = binding
    sys.stderr.write('bench.py rank %d/%d: %s\n' % (rank, world, json.dumps(me)))
    return comm.gather_objects(me, comm.next_tag('rank_report'))


def _pipe_level(level):
    """XFR_PIPE_SLOTS=3 / 2 (tuning / A-B runs, tools/ab_env.sh): three / two forward slots whatever the workload chose (xfr_engine_set_pipeline bit 2)."""
    if level and os.environ.get('XFR_PIPE_SLOTS') == '3':
        return level | 4
    if level and os.environ.get('XFR_PIPE_SLOTS') == '2':
        return level & ~4
    return level


def timed_loop(W, steps, warmup, barrier, world, dev, comm=None):
    """W warm-up steps, then exactly `steps` steps between barrier + synchronize on both sides; the checks on the last step's maps."""
    import torch
    import torch.distributed as dist
    step = W.step
    for _ in range(warmup):
        sal = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        sal = step()
    t_enqueue = time.perf_counter() - t0      # host time to enqueue the K steps (launch-bound if close to dt)
    barrier()
    dt = dt_rank = time.perf_counter() - t0
    cs = chain_stats()                        # process-wide counters right after the timed loop: nothing but product-path steps so far
    if world > 1:
        dt = comm.max_float(dt, dev)
    # every map of the last step: finite, non-negative, unit sum
    ok = bool(torch.isfinite(sal).all().item()) and float(sal.min().item()) >= 0.0 and \
        float((sal.sum(dim=(1, 2)) - 1.0).abs().max().item()) < 1e-3
    return {'dt': dt, 'dt_rank': dt_rank, 't_enqueue': t_enqueue, 'sal': sal, 'ok': ok, 'chain': cs}


def executed_flops(W, reps=2):
    """(GEMM ms, launches, executed GEMM FLOPs) per step of the one-stream schedule: HIP events around every launch."""
    eng = W.eng
    eng.set_profile(True)
    s_ms, s_n, s_fl = 0.0, 0, 0.0
    fam = {'fp32': [0.0, 0, 0.0], 'bf16x6': [0.0, 0, 0.0]}
    for _ in range(reps):
        W.step(False)
        ms, n, fl = eng.get_profile(); s_ms += ms; s_n += n; s_fl += fl
        for k, v in eng.get_profile_by_kernel().items():
            for i in range(3):
                fam[k][i] += v[i]
    eng.set_profile(False)
    W.by_family_serial = {k: (v[0] / reps, v[1] / float(reps), v[2] / reps) for k, v in fam.items()}
    return s_ms / reps, s_n / reps, s_fl / reps


def roofline_object(W, step, ms_step, dev, reps=2, launch_log_out=None):
    """The `roofline` object of one workload (the headline and every secondary get the same one): (1) the one-stream schedule with HIP events
    around every GEMM launch -- what rocprofv3 --kernel-trace reproduces (profiles/rNN/kernel_stats_serial*.csv) -- and the executed GEMM FLOPs;
    (2) the TIMED schedule from the launch log the kernels write themselves (streams overlapped as timed); (3) the shader clock held meanwhile;
    (4) HBM bytes per launch from the committed PMC passes.  launch_log_out: keep the raw launch log of (2) as that CSV file
    (profiles/frac_from_launch_log.py recomputes `frac` from it).  Returns (object, timeline analysis, FLOPs per step used)."""
    from xfr_amd import tuning
    B = W.B
    peak = PEAK_F32_MFMA / 1e12
    s_ms, s_n, s_fl = executed_flops(W, reps)
    # the numerator of every fraction: never more than what the launches executed (ResNets: the stem's backward-data GEMM is --
    # correctly -- never run, so executed < 6 F_fwd; Light-CNN 'affineonly' needs no relu(W) forward)
    alg_step = min(W.flop_per_unit * B, s_fl)
    for _ in range(3):
        step()
    csv = tuning.record_launch_log(step, 6, dev, csv_path=launch_log_out)
    tl = tuning.analyse_launch_log(csv, 6, alg_step)
    if launch_log_out is None:
        os.remove(csv)
    clk = tuning.shader_clock(step, 6, dev)
    achieved = tl['achieved_over_union_TFLOPs']
    roof = {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak, 'traffic': None,
            'kernel': 'conv_gemm kernels (all GEMM launches of one step), timed multi-stream schedule',
            'how': 'min(algorithmic, executed) FLOPs / union of the GEMM launches\' busy intervals (in-kernel s_memrealtime stamps, xfr_amd/tuning.py; '
                   'profiles/frac_from_launch_log.py recomputes it from the committed launch log)',
            'launches_per_step': tl['launches_per_step'], 'gemm_busy_ms_per_step': tl['gemm_union_busy_ms_per_step'],
            'avg_launch_ms': tl['avg_launch_ms_in_union'], 'step_ms_while_logging': tl['ms_per_step'],
            'concurrent_launches_ms_per_step': tl['concurrent_launches_ms_per_step'],
            # the same FLOPs over the TIMED step (every non-GEMM kernel included)
            'frac_timed': alg_step / (ms_step * 1e-3) / PEAK_F32_MFMA, 'achieved_timed': alg_step / (ms_step * 1e-3) / 1e12,
            # one stream (the schedule rocprofv3 --kernel-trace sees): sum of the launch durations, HIP events
            'frac_serial': alg_step / (s_ms * 1e-3) / PEAK_F32_MFMA, 'achieved_serial': alg_step / (s_ms * 1e-3) / 1e12,
            'gemm_ms_per_step_serial': s_ms, 'avg_launch_ms_serial': s_ms / max(s_n, 1),
            'executed_flop_per_step': s_fl, 'algorithmic_flop_per_step': W.flop_per_unit * B, 'flop_per_step_used': alg_step,
            }
    share = tl.get('split_flop_share', 0.0)
    if share > 0:
        # --split-gemm: part of the FLOPs ran as bf16x6 on the bf16 pipe (2516.6 / 6 = 419.4 TFLOP/s fp32-equivalent).  The roofline is then the
        # work-weighted (harmonic) mix of the two peaks; the fraction against the fp32 MFMA peak alone is kept beside it
        mixed = 1.0 / (share / (PEAK_BF16_MFMA / 6e12) + (1.0 - share) / peak)
        roof.update({'peak': mixed, 'frac': achieved / mixed, 'frac_vs_fp32_mfma_peak': achieved / peak, 'bf16x6_flop_share': share,
                     'bf16x6_launches_per_step': tl.get('split_launches_per_step'),
                     'frac_timed': roof['achieved_timed'] / mixed, 'frac_serial': roof['achieved_serial'] / mixed,
                     'peak_note': 'harmonic mix of the fp32 MFMA peak (%.1f) and the bf16x6 ceiling (%.1f fp32-equivalent TFLOP/s = the bf16 MFMA peak / 6) by FLOP '
                                  'share; every frac* of this object is against it' % (peak, PEAK_BF16_MFMA / 6e12)})
    # every kernel family against ITS OWN ceiling, one-stream schedule (HIP events around every launch: what rocprofv3 --kernel-trace reproduces) -- a slow
    # kernel cannot hide in the mixed figure above
    fams = {}
    for k, pk in (('fp32', peak), ('bf16x6', PEAK_BF16_MFMA / 6e12)):
        ms_f, n_f, fl_f = W.by_family_serial.get(k, (0.0, 0, 0.0))
        if n_f > 0:
            fams[k] = {'launches_per_step': n_f, 'executed_flop_per_step': fl_f, 'gemm_ms_per_step_serial': ms_f, 'achieved_serial': fl_f / (ms_f * 1e-3) / 1e12,
                       'peak': pk, 'frac_serial': fl_f / (ms_f * 1e-3) / 1e12 / pk, 'mfma_pipe_ms_at_peak': fl_f / (pk * 1e12) * 1e3}
    roof['by_family'] = fams
    roof['by_family_note'] = ('fp32: v_mfma_f32_32x32x2_f32 kernels against 157.3 TFLOP/s; bf16x6: conv_gemm_split_kernel against 419.4 fp32-equivalent TFLOP/s '
                              '(bf16 MFMA peak / 6 products); mfma_pipe_ms_at_peak = the matrix-pipe time the family\'s FLOPs need at that peak')
    if clk:
        # 157.3 TFLOP/s is the peak at the nominal 2.4 GHz; what the chip can do at the clock it actually held
        roof['shader_clock_GHz'] = clk
        roof['peak_sustained'] = roof['peak'] * clk['p50'] / 2.4       # the roofline above at the clock actually held (nominal 2.4 GHz)
        roof['frac_of_sustained'] = achieved / roof['peak_sustained']
    if W.pmc_tag is not None:
        roof.update(pmc_traffic(ROOT, W.pmc_tag))
        roof.update(pmc_mfma(ROOT, W.pmc_tag, tl['launches_per_step'], ms_step, clk['p50'] if clk else None))
    return roof, tl, alg_step


def run_secondary(model, dev, steps, warmup, chain_before):
    """One secondary BASELINE.json configuration at N = 1 under the main line's rules: resident inputs, W untimed + K timed steps between
    synchronisations, every map checked, row 0 against the reference's map where a fixture exists."""
    import torch

    class A(object):
        pass
    a = A()
    a.model, a.batch, a.mode = model, None, None
    W = make_workload(a, dev, 0)
    W.eng.set_pipeline(_pipe_level(W.pipeline))
    r = timed_loop(W, steps, warmup, torch.cuda.synchronize, 1, dev)
    ms_step = 1e3 * r['dt'] / steps
    ok, row0 = r['ok'], None
    if W.fixture:
        row0 = fixture_cosine(r['sal'][0], W.fixture)
        ok = ok and row0 is not None and row0 >= ROW0_COS
    interp = r['chain'][1] - chain_before[1]
    ok = ok and interp == 0
    roof, _, alg_step = roofline_object(W, W.step, ms_step, dev, reps=1)
    n_launch = roof['launches_per_step']
    out = {'model': model, 'metric': W.metric, 'workload': W.work, 'value': W.B * steps / r['dt'], 'unit': 'maps/s', 'ms_per_step': ms_step,
           'steps': steps, 'warmup': warmup, 'frac_timed': roof['frac_timed'],
           'algorithmic_flop_per_step': alg_step, 'gemm_launches_per_step': n_launch,
           'outputs_ok': ok, 'row0_cosine_vs_reference': row0, 'interpreted_chain_launches': interp, 'roofline': roof}
    W.eng.close()
    del W
    torch.cuda.empty_cache()
    return out


def run_inpainting_game(dev, jobs=64, group=8, mates=4, topk=32, num_classes=65359, workers=2, engine_batch=None):
    """BASELINE.json configs[4] on ONE GPU: the job mix of the reference's inpainting-game generator on ResNet-101 (eval/
    generate_inpaintinggame_wb_saliency_maps_multigpu.py:121-231; python/xfr/inpainting_game/generate_whitebox_saliency.py:79-214) -- per job meanEBP
    over the 65359-way hooked classifier, contrastive and truncated-contrastive triplet EBP from `mates` averaged mate / non-mate encodings, and
    weighted-subtree EBP top-32 ('norelu', whitebox.py:647-737) -- `group` jobs per batch through xfr_amd.inpainting_game.run_jobs_batched (what
    tools/inpainting_game_workload.py --group 8 runs; that tool shards the jobs over ranks).  `workers` job groups are in flight at once: that many
    engines, each on its own host thread and stream -- the reference runs a pool of workers too (one per GPU, ..._multigpu.py:193); here a second worker
    on the SAME GPU fills the launch gaps and the four host synchronisation points per weighted-subtree call of the first (small batches: the device idles
    two thirds of a call).  jobs/s over `jobs` jobs, ms per method (a second pass, ONE worker, device synchronised between methods), and for the weighted
    subtree the GEMM FLOPs its launches executed per probe (in-kernel launch log) over its wall time against the fp32 MFMA peak -- one call at a time
    and with all workers running it."""
    import threading
    import numpy as np
    import torch
    from xfr_amd import inpainting_game as IG, synth, tuning
    from xfr_amd.engine import Engine
    from xfr_amd.models import resnet, whitebox as WB
    bb = resnet.ResNet([3, 4, 23, 3], num_classes=num_classes)
    bb.to(dev)
    sd = synth.synth_state_dict(bb, seed=0)
    wbs, streams = [], []
    for _ in range(workers):
        wbn = WB.WhiteboxSTResnet(bb)
        wbs.append(WB.Whitebox(wbn, ebp_subtree_mode='norelu'))  # eval/create_wbnet.py:51-52 default for resnetv4/v6
        wbn._program = bb.build_program()
        wbn._engine = Engine(wbn._program, engine_batch or max(32, 16 * group), dev)
        wbn._engine_key = (str(bb.device), id(bb))
        wbn._engine.load_weights(sd)
        wbn._engine.loaded_version = bb.version
        streams.append(torch.cuda.Stream(device=dev))
    wb = wbs[0]
    # (no kernel knobs here: launches of fewer than 128 tiles leave the bf16x6 kernel by the library's own rule -- round 5 switched it off by hand)
    k = mates
    pool = [synth.synth_smooth_images(2 * k + 1, (3, 224, 224), seed=10000 + j, mean=resnet.MEAN_RGB).to(dev) for j in range(8)]

    def batch(g0):
        return [(list(pool[j % 8][1:1 + k]), list(pool[j % 8][1 + k:]), pool[j % 8][0]) for j in range(g0, min(jobs, g0 + group))]

    def check(res):
        ok = True
        for maps in res.values():
            for m in maps:
                m = np.asarray(m)
                ok = ok and m.shape == (112, 112) and bool(np.isfinite(m).all()) and abs(float(m.sum()) - 1.0) < 1e-3
        return ok

    def in_workers(fn):
        """fn(w) on worker w's thread and stream; returns the results (an exception of a worker is re-raised here)"""
        out, err = [None] * workers, []

        def body(w):
            try:
                with torch.cuda.device(dev), torch.cuda.stream(streams[w]):
                    out[w] = fn(w)
                    streams[w].synchronize()
            except BaseException as ex:      # noqa: BLE001
                err.append(ex)
        ts = [threading.Thread(target=body, args=(w,)) for w in range(workers)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if err:
            raise err[0]
        return out

    groups = list(range(0, jobs, group))

    def run_all():
        return all(in_workers(lambda w: all([check(IG.run_jobs_batched(wbs[w], batch(g0), 'resnetv4_pytorch', 'norelu', 6, dev, topk=topk)) for g0 in groups[w::workers]])))
    run_all()                                                   # warm: plans, scratch, lazy code objects
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ok = run_all()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # one worker alone, same jobs: what the second worker adds
    t0 = time.perf_counter()
    ok1 = all([check(IG.run_jobs_batched(wb, batch(g0), 'resnetv4_pytorch', 'norelu', 6, dev, topk=topk)) for g0 in groups])
    torch.cuda.synchronize()
    dt1 = time.perf_counter() - t0
    phase = {}
    for g0 in groups:
        IG.run_jobs_batched(wb, batch(g0), 'resnetv4_pytorch', 'norelu', 6, dev, topk=topk, timings=phase)
    # the weighted subtree alone: wall time and executed GEMM FLOPs of one group, one call at a time ...
    ws = ('weighted-subtree',)
    IG.run_jobs_batched(wb, batch(0), 'resnetv4_pytorch', 'norelu', 6, dev, topk=topk, methods=ws)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        IG.run_jobs_batched(wb, batch(0), 'resnetv4_pytorch', 'norelu', 6, dev, topk=topk, methods=ws)
    torch.cuda.synchronize()
    t_ws = (time.perf_counter() - t1) / 3
    # ... and with every worker running it (the job mix's schedule)
    in_workers(lambda w: IG.run_jobs_batched(wbs[w], batch(0), 'resnetv4_pytorch', 'norelu', 6, dev, topk=topk, methods=ws))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    in_workers(lambda w: [IG.run_jobs_batched(wbs[w], batch(0), 'resnetv4_pytorch', 'norelu', 6, dev, topk=topk, methods=ws) for _ in range(3)])
    torch.cuda.synchronize()
    t_ws_conc = (time.perf_counter() - t1) / (3 * workers)
    csv = tuning.record_launch_log(lambda: IG.run_jobs_batched(wb, batch(0), 'resnetv4_pytorch', 'norelu', 6, dev, topk=topk, methods=ws), 0, dev,
                                   launches_per_step_cap=40000)
    fl, n_launch, busy = 0.0, 0, 0
    for ln in list(open(csv))[1:]:
        r = ln.strip().split(',')
        if int(r[9]) > 0 and int(r[10]) > int(r[9]):
            fl += 2.0 * int(r[4]) * int(r[5]) * int(r[2]) * int(r[3])
            busy += int(r[10]) - int(r[9])
            n_launch += 1
    os.remove(csv)
    fl, n_launch, busy_ms = fl / 2, n_launch / 2, busy * 1e-5 / 2           # the log holds two calls
    out = {'model': 'resnet101 inpainting-game job mix', 'metric': 'inpainting-game whitebox saliency jobs/sec, ResNet-101 (4 methods per job), 1 GPU',
           'workload': 'BASELINE.json configs[4] shape on one GPU: %d synthetic jobs, %d mates + %d non-mates per job, %d-way hooked classifier for meanEBP, '
                       'weighted subtree top-%d, mode norelu, %d jobs per batch, %d batches in flight (one engine, host thread and stream each)' % (
                           jobs, k, k, num_classes, topk, group, workers),
           'value': jobs / dt, 'unit': 'jobs/s', 'seconds': dt, 'jobs': jobs, 'workers': workers, 'outputs_ok': bool(ok and ok1),
           'one_worker': {'jobs_s': jobs / dt1, 'seconds': dt1},
           'ms_per_job_by_method': {kk: round(1e3 * v / jobs, 3) for kk, v in phase.items()},
           'weighted_subtree': {'ms_per_probe': 1e3 * t_ws_conc / group, 'ms_per_probe_one_call_at_a_time': 1e3 * t_ws / group, 'probes_per_call': group,
                                'calls_in_flight': workers, 'gemm_launches_per_call': n_launch,
                                'executed_gemm_gflop_per_probe': fl / group / 1e9, 'sum_of_gemm_launch_ms_per_call': busy_ms,
                                'frac_of_peak_over_wall': fl / t_ws_conc / PEAK_F32_MFMA, 'frac_of_peak_over_wall_one_call_at_a_time': fl / t_ws / PEAK_F32_MFMA,
                                'frac_of_peak_while_a_gemm_runs': fl / (busy_ms * 1e-3) / PEAK_F32_MFMA},
           'reference': '~36 h for 541 ResNet-101 jobs on one Titan X (README.md:166) = ~240 s per job'}
    for w_ in wbs:
        w_.net._engine.close()
    torch.cuda.empty_cache()
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (same command line, the environment torch.distributed.run would give them,
    a free rendezvous port on 127.0.0.1), pass rank 0's stdout through -- its ONE JSON line -- and return the largest exit code.  Fewer visible GPUs than
    ranks (a one-GPU box): the ranks share the devices round-robin over gloo -- RCCL cannot put two ranks on one device -- and the line says so."""
    import socket
    import subprocess
    import torch
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ndev = torch.cuda.device_count()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0', XFR_LAUNCHER='bench.py itself (%d ranks spawned, %d device(s) visible)' % (n, ndev))
        if ndev < n:
            env.update(XFR_FORCE_DEVICE=str(r % max(ndev, 1)), XFR_DIST_BACKEND='gloo')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    deadline = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        if deadline is None and any(p.poll() not in (None, 0) for p in procs):
            deadline = time.time() + 120.0            # a rank failed: the others report it (shard.Comm) and leave; do not wait for ever
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
    return max(abs(p.returncode or 0) for p in procs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=None, help='triplets (Light-CNN: images) per GPU per step; default: the BASELINE.json batch of the model')
    ap.add_argument('--mode', default=None)
    ap.add_argument('--model', default='resnet101', choices=['resnet101', 'resnet50_128', 'lightcnn'],
                    help='resnet101 = the BASELINE.json headline; the other two are its secondary configurations')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-whole-host', action='store_true', help='cpu_baseline without the P-process whole-host figure')
    ap.add_argument('--no-secondary', action='store_true', help='N = 1, --model resnet101: do not append the ResNet-50-128d / Light-CNN lines')
    ap.add_argument('--secondary-steps', type=int, default=15)
    ap.add_argument('--no-unfused-ref', action='store_true', help='skip the extra un-fused reference steps of the roofline object')
    ap.add_argument('--no-profile', action='store_true', help='no roofline object (no extra steps after the timed region)')
    ap.add_argument('--serial', action='store_true', help='run every step on ONE stream with per-GEMM HIP events; use under rocprofv3 so that kernel durations are not inflated by concurrent streams')
    ap.add_argument('--no-pipeline', action='store_true', help='do not overlap step i+1 forwards with step i backward')
    ap.add_argument('--profile-csv', default=None, help='with --serial: append one record per GEMM launch to this file (profiles/layer_table.py)')
    ap.add_argument('--launch-log-csv', default=None, help='with --serial: after the timed loop, record 3 more steps with the in-kernel launch log on (one stream) and write it here: per-launch durations from the kernels\' own s_memrealtime stamps (profiles/layer_table.py reads it) -- HIP events misread the first GEMM after an idle queue')
    ap.add_argument('--launch-log-out', default=None, help='keep the raw in-kernel launch log of the TIMED schedule (the one roofline.frac is computed from) as this CSV file; profiles/frac_from_launch_log.py recomputes frac from it')
    ap.add_argument('--timeline-json', default=None, help='write the launch-log analysis of the timed schedule (roofline.timeline) to this file')
    ap.add_argument('--no-sustained', action='store_true', help='skip the >= 10 s sustained loop after the timed region')
    ap.add_argument('--sustained-seconds', type=float, default=10.0)
    ap.add_argument('--fusion', type=int, default=None, help='xfr_engine_set_epilogue_fusion level (default: the library default, 3; 1 leaves BatchNorm / ReLU of the probe forward in their own kernels, 0 un-fuses everything)')
    ap.add_argument('--inpainting-game', action='store_true', help='only the BASELINE.json configs[4] job mix (one GPU): print its object and exit')
    ap.add_argument('--job-workers', type=int, default=2, help='job mix: job groups in flight (one engine, host thread and stream each)')
    ap.add_argument('--job-engine-batch', type=int, default=None, help='job mix: batch capacity of each engine (default 16 x the group of 8 = 128: 32 layerwise sweeps per probe and round; an engine for 128 images is ~95 GB)')
    ap.add_argument('--no-split-leg', action='store_true', help='skip the two other xfr_engine_set_split_gemm modes that ride on the line (split_gemm_modes)')
    ap.add_argument('--split-gemm', type=int, default=None, help='xfr_engine_set_split_gemm(MODE): 0 fp32 MFMA kernels everywhere, 1 bf16x6 forward convolutions of the deep-K layers, 3 the backward-data GEMMs too (the default); + 4: whatever the grid')
    ap.add_argument('--no-lean', action='store_true', help='xfr_engine_set_lean(0): the literal hook operands in every sweep (A/B against the default lean schedule)')
    ap.add_argument('--dry-run', action='store_true', help='rendezvous, weight broadcast, per-rank report, one step, barrier -- then exit (fast failure check on a multi-GPU box)')
    ap.add_argument('--bind', action='store_true', help='pin every rank to its own CPU set (the GPU\'s NUMA node split among the ranks that share it): 8 launch threads of ~500 launches per step each do not migrate or share cores')
    ap.add_argument('--cpu-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-worker-seconds', type=float, default=20.0, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-worker-threads', type=int, default=16, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker_main(args.model, args.cpu_worker_seconds, args.cpu_worker_threads, args.batch, args.mode)

    import torch
    import torch.distributed as dist
    from xfr_amd import shard, tuning

    if os.environ.get('XFR_FAULT_DUMP_S'):       # debugging aid: every thread's Python stack on stderr after that many seconds (and again, repeatedly)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ['XFR_FAULT_DUMP_S']), repeat=True)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ:
        # started like the N = 1 command (`python bench.py --gpus N`), not under torch.distributed.run: be the launcher, like the reference's driver,
        # which takes --gpus 0 1 2 3 and starts its own workers (eval/generate_inpaintinggame_wb_saliency_maps_multigpu.py:125-130, 193-216)
        sys.exit(spawn_ranks(args.gpus))
    # rendezvous with a bounded wait; a collective backend that does not come up is a reported condition, not a crash (shard.Comm)
    try:
        comm = shard.Comm()
    except BaseException:          # noqa: BLE001 -- the side channel itself (TCPStore rendezvous) failed: still one JSON line, still a non-zero exit
        import traceback
        text = traceback.format_exc()
        sys.stderr.write('bench.py rank %s: rendezvous FAILED:\n%s' % (os.environ.get('RANK', '0'), text))
        if os.environ.get('RANK', '0') == '0':
            print(json.dumps({'error': text.strip().splitlines()[-1], 'n_gpus': args.gpus, 'stage': 'rendezvous'}))
        sys.stdout.flush()
        os._exit(4)
    rank, world, local = comm.rank, comm.world, comm.local
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)\n' % (args.gpus, world))
        if args.gpus != 1 or world != 1:
            sys.exit(2)
    try:
        run(args, comm)
    except SystemExit:
        raise
    except BaseException:          # noqa: BLE001 -- every rank's exception text reaches rank 0's output, whatever it was
        import traceback
        text = traceback.format_exc()
        comm.report_error(text)
        sys.stderr.write('bench.py rank %d/%d FAILED:\n%s' % (rank, world, text))
        if rank == 0:
            errs = comm.collect_errors(2.0)
            errs[0] = text
            print(json.dumps({'error': text.strip().splitlines()[-1], 'n_gpus': world, 'rank_errors': {str(k): v for k, v in sorted(errs.items())}}))
        sys.stdout.flush()
        os._exit(4)                # no clean shutdown of a process group whose peers may be gone


def run(args, comm):
    import torch
    from xfr_amd import shard, tuning
    rank, world, local = comm.rank, comm.world, comm.local
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    if world > 1:
        # N ranks share the host's CPUs: the default of one intra-op thread per LOGICAL CPU (256 on these boxes, under a cgroup quota of 16) times N
        # ranks stalls every rank's host-side tensor code for minutes (measured: eight ranks, > 100 s in the input synthesis); give each rank its share
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
        torch.set_num_threads(max(1, host_cpu_budget()['usable'] // max(1, local_world)))
    if os.environ.get('XFR_TEST_RAISE_RANK') == str(rank):
        raise RuntimeError('XFR_TEST_RAISE_RANK: simulated failure of rank %d' % rank)
    if args.inpainting_game:
        if rank == 0:
            print(json.dumps(run_inpainting_game(dev, workers=args.job_workers, engine_batch=args.job_engine_batch)))
        comm.close()
        return
    binding = shard.bind_rank_cpus(local, int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))) if args.bind else None
    W = make_workload(args, dev, rank, comm=comm)
    eng, B = W.eng, W.B
    ranks = rank_report(eng, comm, binding)
    if rank == 0 and len({r.get('arena_checksum48') for r in ranks}) != 1:
        sys.stderr.write('bench.py: the ranks hold different parameter arenas after the broadcast: %s\n' % json.dumps(ranks))
        sys.exit(3)

    def barrier():
        comm.barrier()
        torch.cuda.synchronize()

    if args.dry_run:
        sal = W.step(False)
        barrier()
        if rank == 0:
            print(json.dumps({'dry_run': True, 'n_gpus': world, 'weights_via': comm.weights_via, 'collective_backend_ok': comm.collective_ok or world == 1,
                              'ranks': ranks, 'outputs_finite': bool(torch.isfinite(sal).all().item())}))
        comm.close()
        return

    if args.fusion is not None:
        eng.set_epilogue_fusion(args.fusion)
    if args.no_lean:
        eng.set_lean(False)
    if args.split_gemm is not None:
        eng.set_split_gemm(args.split_gemm)
    if not args.no_pipeline and not args.serial:
        eng.set_pipeline(_pipe_level(W.pipeline))      # inputs are resident and never modified: the pipelining contract holds
    step = W.step
    if args.serial:
        eng.set_profile(True)
        if args.profile_csv:
            eng.profile_csv(args.profile_csv)
    r = timed_loop(W, args.steps, args.warmup, barrier, world, dev, comm)
    dt, t_enqueue, sal, ok = r['dt'], r['t_enqueue'], r['sal'], r['ok']
    chain_timed = r['chain']              # (compiled, interpreted, signatures) through the end of the timed loop
    # the product path runs compiled epilogues only: an interpreted launch inside the timed region fails the line
    ok = ok and (chain_timed[1] == 0 or args.fusion is not None)
    # per-rank rates: a straggler shows up as `spread` (the whole-job value below uses the MAX-over-ranks time)
    rank_rates = shard.gather_rank_rates(B * args.steps / r['dt_rank'], dev, comm)
    row0 = None
    if rank == 0 and W.fixture:
        row0 = fixture_cosine(sal[0], W.fixture)
        ok = ok and row0 is not None and row0 >= ROW0_COS
    if args.serial:
        eng.profile_csv(None)
        if args.launch_log_csv and rank == 0:
            from xfr_amd import tuning as _tuning
            _tuning.record_launch_log(step, 3, dev, args.launch_log_csv)
        eng.set_profile(False)
    # host cost of enqueueing one step on an EMPTY queue (no back-pressure from a full HIP queue): median of 5
    idle = []
    for _ in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        idle.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    host_idle_ms = 1e3 * sorted(idle)[2]
    one_lat = one_triplet_latency(W) if (rank == 0 and world == 1 and not args.serial) else None
    # sustained loop outside the timed region (the K timed steps take well under a second): long enough for an external
    # power / utilisation sampler to see the device busy, and a steady-state (thermally settled) rate
    sustained = None
    if not args.no_sustained:
        n_sus = 0
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < args.sustained_seconds:
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            n_sus += 20
        sustained = {'maps_s': n_sus * B / (time.perf_counter() - t1), 'seconds': time.perf_counter() - t1, 'steps': n_sus}

    # the same step with uint8 inputs arriving from pinned host memory: H2D copy and preprocessing inside the timed region (never `value`)
    u8_e2e = None
    if rank == 0 and world == 1 and getattr(W, 'u8_step', None) is not None and not args.serial:
        for _ in range(3):
            s8 = W.u8_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n8 = max(5, args.steps)
        for _ in range(n8):
            s8 = W.u8_step()
        torch.cuda.synchronize()
        d8 = time.perf_counter() - t1
        ok8 = bool(torch.isfinite(s8).all().item()) and float((s8.sum(dim=(1, 2)) - 1.0).abs().max().item()) < 1e-3
        u8_e2e = {'maps_s': B * n8 / d8, 'ms_per_step': 1e3 * d8 / n8, 'steps': n8, 'outputs_ok': ok8,
                  'what': 'xfr_triplet_contrastive_u8_host: the %d uint8 224x224x3 images of a step copied from pinned host memory (%.1f MB instead of %.1f MB as fp32) by '
                          'the engine (own copy stream, per-slot staging) and preprocessed on the device, both inside the timed region; no residency promise' % (
                              3 * B, 3 * B * 150528 / 1e6, 3 * B * 602112 / 1e6)}
    roof = None
    unfused_leg = None
    ms_step = 1e3 * dt / args.steps
    if not args.no_profile and rank == 0:
        reps = 2
        roof, tl, alg_step = roofline_object(W, step, ms_step, dev, reps=reps, launch_log_out=args.launch_log_out)
        if args.timeline_json:
            json.dump(tl, open(args.timeline_json, 'w'), indent=1)
        # the same launches with the elementwise epilogues un-fused (convolution work only): reference figure for the MFMA
        # kernel by itself; the product path above is the fused one.  Its merged copy / flush chains accumulate into their
        # destination and therefore run INTERPRETED epilogues: counted separately below, never part of the product path's count.
        if not args.no_unfused_ref and W.model == 'resnet101':
            before = chain_stats()
            eng.set_epilogue_fusion(False)
            eng.set_profile(True)
            u_ms, u_n = 0.0, 0
            for _ in range(reps):
                W.step(False)
                ms, n, fl = eng.get_profile(); u_ms += ms; u_n += n
            eng.set_profile(False)
            eng.set_epilogue_fusion(True)
            after = chain_stats()
            unfused_leg = {'compiled_launches': after[0] - before[0], 'interpreted_launches': after[1] - before[1]}
            roof['unfused_epilogues_serial'] = {'achieved': alg_step * reps / (u_ms * 1e-3) / 1e12, 'frac': alg_step * reps / (u_ms * 1e-3) / (roof['peak'] * 1e12),
                                                'gemm_ms_per_step': u_ms / reps, 'avg_launch_ms': u_ms / max(u_n, 1)}

    # the same step in the other two modes of xfr_engine_set_split_gemm (conv_gemm_split.hip K17; the headline runs the default, mode 3: the forward convolutions
    # AND the sweep's backward-data GEMMs of the deep-K layers as bf16x6 on the bf16 matrix pipe).  mode 0: fp32 MFMA kernels everywhere.  mode 1: the forward
    # convolutions only (the round-5 default).  Same output checks as the headline.
    split_leg = None
    if rank == 0 and world == 1 and not args.serial and args.split_gemm is None and not args.no_split_leg:
        split_leg = {}
        for name, mode in (('fp32_mfma_only', 0), ('bf16x6_forward_only', 1)):
            try:
                before = eng.split_gemm_launches()
                eng.set_split_gemm(mode)
                rs = timed_loop(W, args.steps, max(args.warmup, 3), barrier, world, dev, comm)
                n_split = (eng.split_gemm_launches() - before) / float(args.steps + max(args.warmup, 3))
                row0s = fixture_cosine(rs['sal'][0], W.fixture) if W.fixture else None
                split_leg[name] = {'mode': mode, 'maps_s': B * args.steps / rs['dt'], 'ms_per_step': 1e3 * rs['dt'] / args.steps, 'bf16x6_launches_per_step': n_split,
                                   'outputs_finite_and_normalised': bool(rs['ok']), 'row0_cosine_vs_reference': row0s,
                                   'row0_check_passes': (row0s is not None and row0s >= ROW0_COS) if W.fixture else None}
            except Exception as ex:      # these legs never fail the line
                split_leg[name] = {'mode': mode, 'error': repr(ex)}
            finally:
                eng.set_split_gemm(3)

    secondary = None
    if rank == 0 and world == 1 and args.model == 'resnet101' and not args.no_secondary and not args.serial and args.batch is None and args.mode is None:
        # BASELINE.json configs[2] and configs[3] on the same line (the driver only runs the default command)
        chain_main = chain_stats()
        eng.close()
        del eng
        W.eng = None
        torch.cuda.empty_cache()
        secondary = []
        for m in ('resnet50_128', 'lightcnn'):
            try:
                secondary.append(run_secondary(m, dev, args.secondary_steps, 3, chain_stats()))
            except Exception as ex:      # the headline must still be printed
                secondary.append({'model': m, 'error': repr(ex), 'outputs_ok': False})
        try:
            secondary.append(run_inpainting_game(dev, workers=args.job_workers, engine_batch=args.job_engine_batch))
        except Exception as ex:
            secondary.append({'model': 'resnet101 inpainting-game job mix', 'error': repr(ex), 'outputs_ok': False})
        del chain_main

    if rank == 0:
        value = world * B * args.steps / dt
        line = {
            'metric': W.metric, 'value': value, 'unit': 'maps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'dtype_note': 'fp32 storage, operands and accumulation throughout.  GEMMs: v_mfma_f32_32x32x2_f32; the forward convolutions and backward-data GEMMs of the '
                          'deep-K stride-1 layers (xfr_engine_set_split_gemm mode %s) as bf16x6 -- every fp32 operand as the exact sum of three bf16 pieces, six exact piece '
                          'products, partial sums of 48 K-terms folded into fp32 registers with round-to-nearest adds: error against float64 BELOW the fp32 MFMA kernels\' '
                          '(profiles/r6/conv_error_probe.txt; conv_gemm_split.hip K17; split_gemm_modes.fp32_mfma_only is the same step on fp32 MFMAs alone)' % (args.split_gemm if args.split_gemm is not None else 3),
            'config': {'workload': W.work, 'units_per_gpu': B,
                       'parallelism': 'independent triplets, %d process(es), weights broadcast once' % world},
            'outputs_ok': ok, 'row0_cosine_vs_reference': row0,
            # host time per step while the queue is full (back-pressure included) and on an empty queue (the true launch cost)
            'host_enqueue_ms_per_step': 1e3 * t_enqueue / args.steps, 'host_enqueue_idle_ms': host_idle_ms,
            'rank_maps_s': rank_rates,
            'ranks': ranks,
            # how the parameters reached the ranks: 'broadcast' (one RCCL broadcast of the packed arena), 'local_pack_fallback' (the collective
            # backend or the broadcast failed somewhere: every rank packed from the seed; the checksums above still agree), 'local_pack' (N = 1)
            'weights_via': comm.weights_via, 'collective_backend_ok': bool(comm.collective_ok or world == 1),
        }
        errs = comm.collect_errors()
        if errs:
            line['rank_errors'] = {str(k): v for k, v in sorted(errs.items())}
        line['launcher'] = os.environ.get('XFR_LAUNCHER', 'external (torch.distributed.run or the caller\'s environment)' if world > 1 else 'none (one process)')
        if world > 1:
            line['scaling_note'] = 'N > 1 has only ever run with several ranks on ONE GPU here (gloo); no multi-GPU node was available to the builder: no multi-GPU hardware number exists'
        if u8_e2e is not None:
            line['u8_end_to_end'] = u8_e2e
        if one_lat is not None:
            line['one_triplet_latency'] = one_lat
        if split_leg is not None:
            line['split_gemm_modes'] = split_leg
        if sustained is not None:
            line['sustained_maps_s'] = world * sustained['maps_s']
            line['sustained'] = sustained
        # counters as they stood right after the timed loop (warm-up + timed steps: the product path); the un-fused reference leg apart
        line['chain_epilogues'] = {'compiled_launches': chain_timed[0], 'interpreted_launches': chain_timed[1], 'signatures': chain_timed[2],
                                   'when': 'through the end of the timed loop', 'unfused_reference_leg': unfused_leg}
        if roof is not None:
            line['roofline'] = roof
        if secondary is not None:
            line['secondary'] = secondary
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = W.cpu_baseline(20.0, not args.no_whole_host)
        print(json.dumps(line))
        sys.stdout.flush()
    comm.close()


if __name__ == '__main__':
    main()
