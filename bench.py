#!/usr/bin/env python
"""bench.py -- triplet-contrastive-EBP saliency maps/sec, ResNet-101 224x224 (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of B synthetic triplets per GPU, inputs resident in HBM:
    encode(mates[B]); encode(nonmates[B]);                      (demo/test_whitebox.py:71-72)
    classifier rows = encoding / 2500                           (demo/test_whitebox.py:129)
    contrastive_ebp(probes[B], 0, 1) -> B saliency maps 112x112 (whitebox.py:506-527), per-triplet classifier
Workload = BASELINE.json configs[1]: ResNet-101 triplet contrastive EBP, batch=32 synthetic 224x224 triplets per
GPU, ebp_subtree_mode 'affineonly_with_prior' (the demo default), fp32 throughout.

Multi-GPU: one process per GPU (torch.distributed.run), triplets sharded embarrassingly, weights packed on rank 0
and broadcast once over RCCL; no steady-state collective => "scaling": "weak" (B triplets per GPU).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_FWD_R101 = 14.419e9            # 2*MAC over conv+linear, measured on the reference modules
FLOPS_PER_TRIPLET = 6 * F_FWD_R101   # 2 encodes + true fwd + relu(W) fwd + 2 backward-data sweeps = 86.51 GFLOP
PEAK_F32_MFMA = 157.3e12         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD * 1024 SIMD * 2.4 GHz


def cpu_baseline(sd, probes, mates, nonmates, mode, budget_s=20.0):
    """The oracle (kind "port": hook-free CPU restatement of the reference, bit-exact against it in the build
    container) timed on this box's host cores on a bounded sample of the same workload: whole triplets
    (2 encodes + contrastive EBP) until ~budget_s of CPU time is spent (at least one).  The reference's path is batch 1
    (whitebox.py:512); its convolutions stop scaling long before a 128-core host is used up (measured on the GPU box:
    8.7 s per triplet with 128 threads, 1.0 s with 32, 0.46 s with 16), so the thread count is chosen by a short probe and
    reported as `cores`."""
    import torch
    from oracle import ebp_oracle as O
    ow = O.OracleWhitebox('stresnet101', sd, ('hooked', None), mode)

    def triplet(i):
        xm = ow.encode(mates[i:i + 1]) / 2500.0
        xn = ow.encode(nonmates[i:i + 1]) / 2500.0
        ow.set_triplet_classifier(xm, xn)
        ow.contrastive_ebp(probes[i:i + 1], 0, 1)

    ncpu = os.cpu_count() or 1
    best_t, best_dt = None, None
    for th in sorted({min(ncpu, c) for c in (8, 16, 32)}):
        torch.set_num_threads(th)
        t0 = time.time()
        triplet(0)
        d = time.time() - t0
        if best_dt is None or d < best_dt:
            best_t, best_dt = th, d
    torch.set_num_threads(best_t)
    n = 0
    t0 = time.time()
    while True:
        triplet(n % probes.shape[0])
        n += 1
        if time.time() - t0 > budget_s or n >= 64:
            break
    dt = time.time() - t0
    out = {'value': n / dt, 'unit': 'maps/s', 'cores': int(best_t), 'kind': 'port',
           'sample': '%d ResNet-101 triplet(s) (2 encodes + contrastive_ebp each), batch 1, %.1f s, %d threads (best of 8/16/32; host has %d)'
                     % (n, dt, best_t, ncpu)}
    out.update(port_vs_reference(ROOT))
    return out


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


def secondary(args, dev, rank, world):
    """BASELINE.json configs[2] (VGGFace2 ResNet-50-128d truncated contrastive EBP, batch 64, mode norelu) and configs[3]
    (Light-CNN-29v2 plain EBP, batch 128, 80013-way hooked classifier, mode affineonly).  Same timing contract."""
    import torch
    from xfr_amd import synth
    from xfr_amd.engine import Engine
    from xfr_amd.models import lightcnn, resnet50_128
    if args.model == 'resnet50_128':
        B = args.batch if args.batch != 32 else 64
        mode = args.mode or 'norelu'
        bb = resnet50_128.Resnet50_128()
        flops = 6 * 7.712e9
        prog = bb.build_program()
        eng = Engine(prog, 2 * B, dev)
        eng.load_weights(synth.synth_state_dict(bb, seed=0))
        eng.set_mode(mode)
        eng.set_pipeline(True)
        imgs = synth.bench_images(B, (3, 224, 224), seed=1234 + rank, mean=(131.0912, 103.8827, 91.4953)).to(dev)
        gallery, probes = imgs[:2 * B].contiguous(), imgs[2 * B:].contiguous()
        enc_t = prog.marks['encode']
        step = lambda: eng.triplet_contrastive(probes, gallery, enc_t, 1.0 / 2500.0, 20.0, inputs_ready=True)   # noqa: E731
        metric = 'triplet truncated-contrastive-EBP (20 %) saliency maps/sec, VGGFace2 ResNet-50-128d 224x224'
        work = 'ResNet-50-128d truncated contrastive EBP, batch=%d synthetic triplets per GPU, mode %s' % (B, mode)
    else:
        B = args.batch if args.batch != 32 else 128
        mode = args.mode or 'affineonly'
        bb = lightcnn.LightCNN_29Layers_v2(num_classes=80013)
        flops = 3 * 7.234e9
        prog = bb.build_program()
        eng = Engine(prog, B, dev)
        eng.load_weights(synth.synth_state_dict(bb, seed=0))
        eng.set_mode(mode)
        eng.set_pipeline(2)          # forward of step i+1 under the backward of step i; x is resident (inputs_ready below)
        x = synth.synth_images(B, (1, 128, 128), seed=1234 + rank, scale255=False).to(dev)
        seed = torch.zeros((1, B, 80013), device=dev)
        seed[0, :, 0] = 1.0
        cls_t = prog.marks['classify']

        def step():
            _, pooled = eng.ebp(x, cls_t, seed, want_mwp=False, want_pooled=True, inputs_ready=True)
            return eng.mwp_to_saliency(pooled[0])
        metric = 'EBP saliency maps/sec, Light-CNN-29v2 128x128 (80013-way hooked classifier)'
        work = 'Light-CNN-29v2 excitation backprop, batch=%d synthetic images per GPU, mode %s' % (B, mode)
    for _ in range(args.warmup):
        sal = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sal = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = bool(torch.isfinite(sal).all().item()) and float((sal.sum(dim=(1, 2)) - 1.0).abs().max().item()) < 1e-3
    row0 = None
    if args.model == 'resnet50_128' and B == 64 and rank == 0:
        row0 = fixture_cosine(sal[0], 'bench/r50')
        ok = ok and row0 is not None and row0 >= ROW0_COS
    eng.set_profile(True)
    step()
    ms, nl, fl = eng.get_profile()
    eng.set_profile(False)
    if rank == 0:
        # algorithmic work of THIS mode: the engine only runs the relu(W) forward where a hook divides by X (Light-CNN in
        # 'affineonly' needs none), so use the executed GEMM FLOPs (== algorithmic minimum for the mode), capped by 3 F_fwd
        ach = min(fl, flops * B) / (ms * 1e-3) / 1e12
        print(json.dumps({'metric': metric, 'value': world * B * args.steps / dt, 'unit': 'maps/s', 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                          'config': {'workload': work}, 'outputs_ok': ok, 'row0_cosine_vs_reference': row0,
                          'roofline': {'bound': 'mfma', 'achieved': ach, 'peak': PEAK_F32_MFMA / 1e12, 'unit': 'TFLOP/s',
                                       'frac': ach / (PEAK_F32_MFMA / 1e12), 'traffic': None, 'launches_per_step': nl,
                                       'gemm_ms_per_step': ms}}))


ROW0_COS = 0.999     # sample 0 against the map the reference computes for the same triplet (tests/golden/golden_bench.npz)


def fixture_cosine(sal0, key):
    """Cosine between the engine's map for sample 0 of rank 0's batch and the committed reference map of that triplet."""
    import numpy as np
    f = os.path.join(ROOT, 'tests', 'golden', 'golden_bench.npz')
    if not os.path.exists(f):
        return None
    want = np.load(f)[key + '/map'].astype(np.float64).ravel()
    got = sal0.detach().cpu().numpy().astype(np.float64).ravel()
    return float(got @ want / max(np.linalg.norm(got) * np.linalg.norm(want), 1e-300))


def chain_stats():
    import ctypes
    from xfr_amd import _lib
    c, i, n = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32()
    _lib.check(_lib.load().xfr_chain_epilogue_stats(ctypes.byref(c), ctypes.byref(i), ctypes.byref(n)))
    return c.value, i.value, n.value


def pmc_traffic(root):
    """HBM bytes per conv_gemm launch from the committed PMC passes of this same command (profiles/rNN/pmc_FETCH_SIZE.txt,
    pmc_WRITE_SIZE.txt: separate rocprofv3 --pmc runs, KB summed over the dispatches of 3 steps).  FETCH_SIZE is doubled:
    on gfx950 it reports half the bytes of 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM section); the dword-wide
    im2col reads of the KxK layers are not calibrated, so this is an upper bound.  Counters cannot be read from inside
    the timed process, hence the file; {} if no profile has been committed."""
    import glob
    import re
    dirs = sorted(glob.glob(os.path.join(root, 'profiles', 'r[0-9]*')))
    for d in reversed(dirs):
        try:
            vals = {}
            for name in ('FETCH_SIZE', 'WRITE_SIZE'):
                for ln in open(os.path.join(d, 'pmc_%s.txt' % name)):
                    m = re.match(r'conv_gemm_kernel\s+dispatches\s+(\d+)\s+.*%s=([0-9.e+]+)' % name, ln)
                    if m:
                        vals[name] = (int(m.group(1)), float(m.group(2)))
            (n, f), (n2, w) = vals['FETCH_SIZE'], vals['WRITE_SIZE']
            return {'traffic': (2.0 * f * 1024 / n + w * 1024 / n2), 'traffic_unit': 'HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)',
                    'traffic_source': os.path.relpath(d, root) + '/pmc_FETCH_SIZE.txt, pmc_WRITE_SIZE.txt'}
        except (OSError, KeyError, ValueError):
            continue
    return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='triplets per GPU per step')
    ap.add_argument('--mode', default=None)
    ap.add_argument('--model', default='resnet101', choices=['resnet101', 'resnet50_128', 'lightcnn'],
                    help='resnet101 = the BASELINE.json headline; the other two are its secondary configurations')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-unfused-ref', action='store_true', help='skip the extra un-fused reference steps of the roofline object (used under rocprofv3)')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--serial', action='store_true', help='run every step on ONE stream with per-GEMM HIP events (what the roofline figure is measured on); use under rocprofv3 so that kernel durations are not inflated by concurrent streams')
    ap.add_argument('--no-pipeline', action='store_true', help='do not overlap step i+1 forwards with step i backward')
    ap.add_argument('--lanes', type=int, default=1, help='engines that take consecutive steps in turn, each on its own stream (xfr_amd.engine.EngineLanes); 1 = one engine')
    ap.add_argument('--profile-csv', default=None, help='with --serial: append one record per GEMM launch to this file (profiles/layer_table.py)')
    ap.add_argument('--no-sustained', action='store_true', help='skip the >= 10 s sustained loop after the timed region')
    ap.add_argument('--sustained-seconds', type=float, default=10.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from xfr_amd import shard, synth
    from xfr_amd.models import resnet, whitebox as WB

    rank, world, local = shard.init_process_group()
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)\n' % (args.gpus, world))
        if args.gpus != 1 or world != 1:
            sys.exit(2)
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    if args.model != 'resnet101':
        return secondary(args, dev, rank, world)
    if args.mode is None:
        args.mode = 'affineonly_with_prior'
    B = args.batch

    bb = resnet.ResNet([3, 4, 23, 3], num_classes=2)   # fc2 is replaced by the per-triplet classifier anyway
    bb.to(dev)
    wbn = WB.WhiteboxSTResnet(bb)
    wbn.default_max_batch = B
    wb = WB.Whitebox(wbn, ebp_subtree_mode=args.mode)
    sd_holder = {}

    def make_sd():
        sd_holder['sd'] = synth.synth_state_dict(bb, seed=0, recipe='mild')
        return sd_holder['sd']
    # engine without weights; rank 0 packs, everybody receives the arena over RCCL
    from xfr_amd.engine import Engine
    wbn._program = bb.build_program()
    wbn._engine = Engine(wbn._program, 2 * B, dev)    # the two encode batches of a step run as one 2B-image forward
    wbn._engine_key = (str(bb.device), id(bb))
    shard.load_and_broadcast(wbn._engine, make_sd, src=0)
    wbn._engine.loaded_version = bb.version
    eng = wb._engine(B)
    enc_t = wbn._program.marks['encode']
    if not args.no_pipeline:
        eng.set_pipeline(True)      # inputs are resident and never modified: the pipelining contract holds

    # synthetic triplets of this rank's shard, resident in HBM (uint8-valued ~U[0,255] minus the RGB mean)
    lo = rank * B
    imgs = synth.bench_images(B, (3, 224, 224), seed=1234 + rank, mean=resnet.MEAN_RGB)
    mates, nonmates, probes = imgs[0:B].to(dev), imgs[B:2 * B].to(dev), imgs[2 * B:3 * B].to(dev)
    gallery = torch.cat((mates, nonmates), dim=0)      # [2B,3,224,224] resident in HBM

    lanes = None
    if args.lanes > 1 and not args.serial and not args.no_pipeline:
        # consecutive steps go to independent engines on their own streams: step i+1 fills the ramps of step i
        from xfr_amd.engine import EngineLanes
        lanes = EngineLanes.__new__(EngineLanes)
        others = [Engine(wbn._program, 2 * B, dev) for _ in range(args.lanes - 1)]
        lanes.engines = [eng] + others
        lanes.device, lanes.program, lanes.max_batch, lanes._next = eng.device, wbn._program, 2 * B, 0
        lanes.streams = [torch.cuda.Stream(device=dev) for _ in lanes.engines]
        lanes.share_weights()
        for e2 in others:
            e2.set_mode(args.mode)
            e2.set_pipeline(True)

    def step():
        # encode(mates), encode(nonmates); set_triplet_classifier(x_mate/2500, x_nonmate/2500); contrastive_ebp(probe,0,1)
        if lanes is not None:
            return lanes.triplet_contrastive(probes, gallery, enc_t, 1.0 / 2500.0, None, inputs_ready=True)
        return eng.triplet_contrastive(probes, gallery, enc_t, 1.0 / 2500.0, None, inputs_ready=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.serial:
        eng.set_profile(True)
        if args.profile_csv:
            eng.profile_csv(args.profile_csv)
    for _ in range(args.warmup):
        sal = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sal = step()
    t_enqueue = time.perf_counter() - t0      # host time to enqueue the K steps (launch-bound if close to dt)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # every map of the last step: finite, non-negative, unit sum; sample 0 (rank 0) against the reference's map of that triplet
    ok = bool(torch.isfinite(sal).all().item()) and float(sal.min().item()) >= 0.0 and \
        float((sal.sum(dim=(1, 2)) - 1.0).abs().max().item()) < 1e-3
    row0 = None
    if rank == 0 and B == 32 and args.mode == 'affineonly_with_prior':
        row0 = fixture_cosine(sal[0], 'bench/r101')
        ok = ok and row0 is not None and row0 >= ROW0_COS

    if args.serial:
        eng.set_profile(False)
        eng.profile_csv(None)
    # host cost of enqueueing one step on an EMPTY queue (no back-pressure from a full HIP queue): median of 5
    idle = []
    for _ in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        idle.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    host_idle_ms = 1e3 * sorted(idle)[2]
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
    roof = None
    if not args.no_profile:
        # live GEMM timing: HIP events around every conv_gemm launch on the launch stream, separate steps after the
        # timed region
        eng.set_profile(True)
        tot_ms, tot_n, tot_fl = 0.0, 0, 0.0
        reps = 2
        for _ in range(reps):
            eng.triplet_contrastive(probes, gallery, enc_t, 1.0 / 2500.0, None)   # serialised on one stream while profiling
            ms, n, fl = eng.get_profile(); tot_ms += ms; tot_n += n; tot_fl += fl
        eng.set_profile(False)
        alg = FLOPS_PER_TRIPLET * B * reps
        achieved = alg / (tot_ms * 1e-3) / 1e12
        frac_timed = FLOPS_PER_TRIPLET * B / (dt / args.steps) / PEAK_F32_MFMA
        roof = {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F32_MFMA / 1e12, 'unit': 'TFLOP/s',
                'frac': achieved / (PEAK_F32_MFMA / 1e12), 'traffic': None,
                # the same algorithmic FLOPs over the TIMED step (three streams overlapped, every non-GEMM kernel included)
                'frac_timed': frac_timed, 'achieved_timed': frac_timed * PEAK_F32_MFMA / 1e12,
                'kernel': 'conv_gemm_kernel (all shapes of one step)', 'launches_per_step': tot_n // reps,
                'avg_launch_ms': tot_ms / max(tot_n, 1), 'gemm_ms_per_step': tot_ms / reps,
                'executed_flop_per_step': tot_fl / reps, 'algorithmic_flop_per_step': alg / reps}
        if B == 32 and args.mode == 'affineonly_with_prior':
            roof.update(pmc_traffic(os.path.dirname(os.path.abspath(__file__))))
        # the same launches with the elementwise epilogues un-fused (convolution work only): reference figure for the MFMA
        # kernel by itself; the product path above is the fused one
        if not args.no_unfused_ref:
            eng.set_epilogue_fusion(False)
            eng.set_profile(True)
            u_ms, u_n = 0.0, 0
            for _ in range(reps):
                eng.triplet_contrastive(probes, gallery, enc_t, 1.0 / 2500.0, None)
                ms, n, fl = eng.get_profile(); u_ms += ms; u_n += n
            eng.set_profile(False)
            eng.set_epilogue_fusion(True)
            roof['unfused_epilogues'] = {'achieved': alg / (u_ms * 1e-3) / 1e12, 'frac': alg / (u_ms * 1e-3) / PEAK_F32_MFMA,
                                         'gemm_ms_per_step': u_ms / reps, 'avg_launch_ms': u_ms / max(u_n, 1)}

    if rank == 0:
        value = world * B * args.steps / dt
        line = {
            'metric': 'triplet-contrastive-EBP saliency maps/sec, ResNet-101 224x224',
            'value': value, 'unit': 'maps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'ResNet-101 triplet contrastive EBP, batch=%d synthetic 224x224 triplets per GPU '
                                   '(2 encodes + contrastive_ebp per triplet), mode %s, eps 1e-16' % (B, args.mode),
                       'triplets_per_gpu': B, 'parallelism': 'independent triplets, %d process(es), weights broadcast once' % world},
            'outputs_ok': ok, 'row0_cosine_vs_reference': row0,
            # host time per step while the queue is full (back-pressure included) and on an empty queue (the true launch cost)
            'host_enqueue_ms_per_step': 1e3 * t_enqueue / args.steps, 'host_enqueue_idle_ms': host_idle_ms,
        }
        if sustained is not None:
            line['sustained_maps_s'] = world * sustained['maps_s']
            line['sustained'] = sustained
        cs = chain_stats()
        line['chain_epilogues'] = {'compiled_launches': cs[0], 'interpreted_launches': cs[1], 'signatures': cs[2]}
        if roof is not None:
            line['roofline'] = roof
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(sd_holder['sd'], probes.cpu(), mates.cpu(), nonmates.cpu(), args.mode)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
