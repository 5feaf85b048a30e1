#!/usr/bin/env python
"""BASELINE.json configs[4]: the workload SHAPE of the reference's inpainting-game whitebox saliency generator
(eval/generate_inpaintinggame_wb_saliency_maps_multigpu.py:121-231 + python/xfr/inpainting_game/generate_whitebox_saliency.py:
79-115,207-214,119-205) on synthetic data, sharded over ranks like the reference's one-worker-per-GPU pool.

Per job (= one (subject, mask, probe) of the reference's cartesian job list) the four saliency methods are produced:
    meanEBP                      ebp(probe, ones)                              generate_whitebox_saliency.py:207-214
    contrastive triplet EBP      k mates / k non-mates encoded, averaged, unit-normalised, /2500  -> contrastive_ebp   :79-115
    truncated contrastive (20%)  same classifier                                                                         :106-112
    weighted subtree EBP, top-32 mode 'norelu'                                                                            :119-205
The reference needs ~36 h for its 541 ResNet-101 jobs on one Titan X (README.md:166).  Launch:
    python tools/inpainting_game_workload.py --jobs 64
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 tools/inpainting_game_workload.py --jobs 541
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--jobs', type=int, default=32)
    ap.add_argument('--mates', type=int, default=4, help='mate / non-mate images per job')
    ap.add_argument('--topk', type=int, default=32)
    ap.add_argument('--num-classes', type=int, default=65359)
    ap.add_argument('--output-dir', default=None,
                    help='write {mask_id}-{method}-saliency.npz + overlay PNG per job and method (show.py:196-232); '
                         'methods whose files exist are skipped unless --overwrite (the generator\'s resume)')
    ap.add_argument('--overwrite', action='store_true')
    ap.add_argument('--group', type=int, default=1,
                    help='jobs processed together in shared launches (xfr_amd.inpainting_game.run_jobs_batched); 1 = job by job '
                         'through the reference-shaped callers')
    ap.add_argument('--max-batch', type=int, default=0, help='engine batch capacity (default 32, or 16 * group in group mode: 32 layerwise sweeps per probe and round)')
    ap.add_argument('--phases', action='store_true', help='group mode: time every method separately (adds device synchronisations)')
    ap.add_argument('--workers', type=int, default=1,
                    help='group mode: job groups in flight per GPU -- that many engines, each on its own host thread and stream; a second one fills the launch '
                         'gaps and host synchronisation points of the first (bench.py: 82 -> 121 jobs/s; an engine for 128 images is ~95 GB of workspace)')
    ap.add_argument('--numpy-inputs', action='store_true', help='uint8 H x W x 3 images through convert_from_numpy (PIL) per call, like the reference')
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    from xfr_amd import inpainting_game as IG, saliency_io as SIO, shard, synth
    from xfr_amd.models import resnet, whitebox as WB

    rank, world, local = shard.init_process_group()
    if world > 1:      # N ranks share the host's CPUs (see bench.py): one intra-op thread per logical CPU per rank stalls all of them
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16)) // max(1, int(os.environ.get('LOCAL_WORLD_SIZE', str(world))))))
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    bb = resnet.ResNet([3, 4, 23, 3], num_classes=args.num_classes)
    bb.to(dev)
    wbn = WB.WhiteboxSTResnet(bb)
    wb = WB.Whitebox(wbn, ebp_subtree_mode='norelu')            # eval/create_wbnet.py:51-52 default for resnetv4/v6
    from xfr_amd.engine import Engine
    wbn._program = bb.build_program()
    wbn._engine = Engine(wbn._program, args.max_batch or (32 if args.group <= 1 else max(32, 16 * args.group)), dev)
    wbn._engine_key = (str(bb.device), id(bb))
    packed = {'n': 0}

    def make_sd():
        packed['n'] += 1
        return synth.synth_state_dict(bb, seed=0)
    shard.load_and_broadcast(wbn._engine, make_sd, src=0)     # one load, one broadcast
    wbn._engine.loaded_version = bb.version
    # further workers of this rank (group mode): their own engine and stream, the arena copied on the device from the first one's
    wbs, streams = [wb], [torch.cuda.Stream(device=dev)]
    for _ in range(1, max(1, args.workers) if args.group > 1 else 1):
        wn = WB.WhiteboxSTResnet(bb)
        wbs.append(WB.Whitebox(wn, ebp_subtree_mode='norelu'))
        wn._program = wbn._program
        wn._engine = Engine(wn._program, wbn._engine.max_batch, dev)
        wn._engine_key = wbn._engine_key
        wn._engine.weight_arena().copy_(wbn._engine.weight_arena())
        wn._engine.mark_weights_loaded()
        wn._engine.loaded_version = bb.version
        streams.append(torch.cuda.Stream(device=dev))

    lo, hi = shard.shard_range(args.jobs, rank, world)
    k = args.mates
    done, t_methods, written, probe_u8 = 0, np.zeros(4), 0, None
    # synthetic stand-ins for the aligned IJB-C crops (git-LFS pointers in the reference): a small pool, generated up front.
    # Default: device-resident network-format tensors (Whitebox.convert_from_numpy passes tensors through); --numpy-inputs hands
    # the callers uint8 H x W x 3 arrays like the reference's image_loader does, so PIL preprocessing and the host-to-device
    # copies are inside the timed loop.
    pool = [synth.synth_smooth_images(2 * k + 1, (3, 224, 224), seed=10000 + j, mean=resnet.MEAN_RGB) for j in range(8)]
    if args.numpy_inputs:
        mean = np.asarray(resnet.MEAN_RGB, dtype=np.float32).reshape(1, 3, 1, 1)
        pool = [np.clip(p.numpy() + mean, 0, 255).astype(np.uint8).transpose(0, 2, 3, 1) for p in pool]
    else:
        pool = [p.to(dev) for p in pool]
    mode = wb.ebp_subtree_mode()
    names = {'meanEBP': SIO.method_name('meanEBP', mode, 6, 'cuda'), 'contrastive': SIO.method_name('contrastive', mode, 6, 'cuda'),
             'truncated': SIO.method_name('contrastive', mode, 6, 'cuda', truncate_percent=20),
             'weighted-subtree': SIO.method_name('weighted-subtree', mode, 6, 'cuda', topk=args.topk, mode_weighted='norelu')}

    def displayable(probe):      # stand-in for the aligned crop the overlay is drawn on: min-max scaled to [0, 1]
        disp = (probe.astype(np.float64) if args.numpy_inputs else probe.permute(1, 2, 0).cpu().numpy().astype(np.float64))
        return (disp - disp.min()) / (disp.max() - disp.min() + 1e-9)

    phase = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    import threading
    lock = threading.Lock()
    counts = {'written': 0, 'done': 0, 't': 0.0}

    def run_group(w, g0):
        # group mode: the jobs [g0, g1) whose outputs are not all on disk yet run as ONE batch through every method
        g1 = min(hi, g0 + args.group)
        todo = []
        for job in range(g0, g1):
            odir = os.path.join(args.output_dir, 'subject_ID_%d' % (job % len(pool))) if args.output_dir else None
            missing = args.overwrite or odir is None or not all(all(os.path.exists(f) for f in SIO.saliency_paths(odir, '%05d' % job, nm))
                                                                for nm in names.values())
            if missing:
                todo.append((job, odir))
        wr, dt_g = 0, 0.0
        if todo:
            jobs = []
            for job, _ in todo:
                imgs = pool[job % len(pool)]
                jobs.append((list(imgs[1:1 + k]), list(imgs[1 + k:]), imgs[0]))
            t1 = time.perf_counter()
            res = IG.run_jobs_batched(wbs[w], jobs, 'resnetv4_pytorch', 'norelu', 6, dev, topk=args.topk,
                                      timings=phase if (args.phases and len(wbs) == 1) else None)
            torch.cuda.current_stream(dev).synchronize()
            dt_g = time.perf_counter() - t1
            for i, (job, odir) in enumerate(todo):
                for key, nm in names.items():
                    m = np.asarray(res[key][i])
                    assert m.shape == (112, 112) and np.isfinite(m).all() and abs(float(m.sum()) - 1.0) < 1e-3
                    if odir:
                        wr += int(SIO.create_save_smap(nm, odir, True, lambda m=m: m, '%05d' % job, displayable(jobs[i][2])))
        with lock:
            counts['written'] += wr
            counts['done'] += g1 - g0
            counts['t'] += dt_g

    groups = list(range(lo, hi if args.group > 1 else lo, args.group))
    if len(wbs) == 1:
        for g0 in groups:
            run_group(0, g0)
    else:
        errs = []

        def body(w):
            try:
                with torch.cuda.device(dev), torch.cuda.stream(streams[w]):
                    for g0 in groups[w::len(wbs)]:
                        run_group(w, g0)
            except BaseException as ex:      # noqa: BLE001
                errs.append(ex)
        ts = [threading.Thread(target=body, args=(w,)) for w in range(len(wbs))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]
    done += counts['done']
    written += counts['written']
    t_methods[3] += counts['t']            # group mode reports the per-job total in the last slot (summed over the workers)
    for job in range(lo, hi if args.group <= 1 else lo):
        imgs = pool[job % len(pool)]
        probe, mates, nonmates = imgs[0], list(imgs[1:1 + k]), list(imgs[1 + k:])
        net_name, ver = 'resnetv4_pytorch', 6

        def f_mean():
            wbn._classifier = None                              # the checkpoint's hooked 65359-way fc2 (a fresh wb per job in the reference)
            return IG.mean_ebp(wb, probe, net_name, ver, dev)

        def f_con():
            return IG.run_contrastive_triplet_ebp(wb, mates, nonmates, probe, net_name, ver, None, dev)

        def f_tru():
            return IG.run_contrastive_triplet_ebp(wb, mates, nonmates, probe, net_name, ver, 20, dev)

        def f_sub():
            return IG.run_weighted_subtree_triplet_ebp(wb, mates, nonmates, probe, net_name, 'norelu', ver, dev, topk=args.topk)

        methods = [(names['meanEBP'], f_mean), (names['contrastive'], f_con), (names['truncated'], f_tru), (names['weighted-subtree'], f_sub)]
        stamps = [time.perf_counter()]
        for name, fn in methods:
            def checked(fn=fn):
                m = fn()
                assert m.shape == (112, 112) and np.isfinite(m).all() and abs(float(m.sum()) - 1.0) < 1e-3
                return m
            if args.output_dir:
                if probe_u8 is None:
                    probe_u8 = displayable(probe)
                written += int(SIO.create_save_smap(name, os.path.join(args.output_dir, 'subject_ID_%d' % (job % len(pool))), args.overwrite,
                                                    checked, '%05d' % job, probe_u8))
            else:
                checked()
            stamps.append(time.perf_counter())
        probe_u8 = None
        t, t1, t2, t3, t4 = stamps
        t_methods += np.array([t1 - t, t2 - t1, t3 - t2, t4 - t3])
        done += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = torch.tensor([dt, float(done)], device=dev, dtype=torch.float64)
    if world > 1:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        dt, total = float(tmax[0]), int(stats[1])
    else:
        total = done
    if args.output_dir:        # what this rank did, for the multi-rank test: rank 0 alone packs the parameters
        os.makedirs(args.output_dir, exist_ok=True)
        with open(os.path.join(args.output_dir, 'rank%d.json' % rank), 'w') as f:
            json.dump({'rank': rank, 'world': world, 'jobs': [lo, hi], 'packed_weights': bool(packed['n']), 'maps_written': written}, f)
    if rank == 0:
        per = (t_methods / max(done, 1) * 1e3).round(1).tolist()
        print(json.dumps({'workload': 'inpainting-game whitebox saliency generation shape, ResNet-101, synthetic', 'jobs': total,
                          'n_gpus': world, 'seconds': dt, 'jobs_per_s': total / dt, 'group': args.group, 'workers': len(wbs),
                          'ms_per_job_by_phase_rank0': {k_: round(1e3 * v / max(done, 1), 2) for k_, v in phase.items()} or None,
                          'ms_per_job_rank0': {'meanEBP': per[0], 'contrastive(+%d encodes)' % (2 * k): per[1], 'truncated': per[2],
                                               'weighted_subtree_top%d' % args.topk: per[3]},
                          'maps_written_rank0': written if args.output_dir else None,
                          'reference': '~36 h for 541 jobs on one Titan X (README.md:166) = ~240 s/job'}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
