#!/usr/bin/env python
"""the forward-only `embeddings` sweep of the blackbox / inpainting-game code (python/xfr/models/
blackbox.py:366-414 scores ~6500 RISE-masked copies of one probe; whitebox.py:747-785 is the batched encode it calls) on
synthetic data: N masked probes, encoded in batches that stay in HBM.

* the masks of batch i+1 are generated on a side stream while batch i is encoded (two input buffers); nothing returns to the
  host until the last batch is enqueued;
* parity: 8 masked probes taken at random positions of full 128-image batches are compared with the CPU oracle's encode of the
  same images (batch 1, like the reference);
* prints one JSON line: images/s, forward TFLOP/s (F_fwd = 14.419 GFLOP per image) and its fraction of the fp32 MFMA peak.

    python tools/embeddings_sweep.py --masks 6500 --batch 128
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
    ap.add_argument('--masks', type=int, default=6500)
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--cells', type=int, default=7, help='RISE grid: cells x cells random mask, bilinearly upsampled')
    ap.add_argument('--parity', type=int, default=8, help='masked probes checked against the CPU oracle (0: skip)')
    ap.add_argument('--ab', type=int, default=0, help='A/B in one process: run the sweep this many times with the forward split on and off, alternating, and print one line per run')
    ap.add_argument('--repeat', type=int, default=3, help='run the whole sweep this many times in the process and report the median (a RISE job runs for minutes; '
                    'the first ~2 s of a process -- streams, clocks -- read 5-25 %% low); every sweep is listed in sweeps_images_per_s')
    ap.add_argument('--no-split', action='store_true', help='xfr_engine_set_forward_split(0): one forward per batch on the caller\'s stream (round 3)')
    args = ap.parse_args()
    import numpy as np
    import torch
    from xfr_amd import synth
    from xfr_amd.models import resnet, whitebox as WB

    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    bb = resnet.ResNet([3, 4, 23, 3], num_classes=2)
    sd = synth.synth_state_dict(bb, seed=0)
    bb.load_state_dict(sd)
    bb.to(dev)
    wbn = WB.WhiteboxSTResnet(bb)
    wbn.default_max_batch = args.batch
    wb = WB.Whitebox(wbn)
    wb.batch_size = args.batch
    probe = synth.synth_smooth_images(1, (3, 224, 224), seed=7, mean=resnet.MEAN_RGB).to(dev)
    g = torch.Generator(device='cpu').manual_seed(0)
    n_batches = (args.masks + args.batch - 1) // args.batch
    grids = [(torch.rand((min(args.batch, args.masks - i * args.batch), 1, args.cells, args.cells), generator=g) < 0.5).float().pin_memory()
             for i in range(n_batches)]
    bufs = [torch.empty((args.batch, 3, 224, 224), device=dev) for _ in range(2)]
    gen = torch.cuda.Stream(device=dev)
    ready = [torch.cuda.Event() for _ in range(2)]
    free = [torch.cuda.Event() for _ in range(2)]

    def generate(i):
        """Masked probes of batch i into buffer i % 2, on the side stream."""
        k, n = i % 2, grids[i].shape[0]
        with torch.cuda.stream(gen):
            gen.wait_event(free[k])                      # the encode that last read this buffer is done
            m = torch.nn.functional.interpolate(grids[i].to(dev, non_blocking=True), size=(224, 224), mode='bilinear', align_corners=False)
            torch.mul(probe, m, out=bufs[k][:n])
            ready[k].record(gen)

    main_s = torch.cuda.current_stream(dev)
    for k in range(2):
        free[k].record(main_s)
    ref = wb.encode(probe)
    for _ in range(12):                                  # warm-up: engine at this batch size, clocks up (the first sweep of a process reads 5-25 % low)
        wb.encode(bufs[0])
    if args.no_split:
        wb._engine(args.batch).set_forward_split(False)
    picks = {}
    if args.parity:
        rng = np.random.RandomState(1)
        full = [i for i in range(n_batches) if grids[i].shape[0] == args.batch] or [0]
        for _ in range(args.parity):
            picks.setdefault(int(rng.choice(full)), []).append(int(rng.randint(0, grids[full[0]].shape[0])))
    kept = {}
    if args.ab:
        eng = wb._engine(args.batch)
        for rep in range(2 * args.ab):
            on = rep % 2 == 0
            eng.set_forward_split(on)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            generate(0)
            for i in range(n_batches):
                k, n = i % 2, grids[i].shape[0]
                if i + 1 < n_batches:
                    generate(i + 1)
                main_s.wait_event(ready[k])
                wb.encode(bufs[k][:n])
                free[k].record(main_s)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(json.dumps({'forward_split': on, 'images_per_s': args.masks / dt, 'frac_of_fp32_mfma_peak': args.masks * 14.419e9 / dt / 157.3e12}), flush=True)
        return
    times = []
    for rep in range(max(1, args.repeat)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        generate(0)
        sims = []
        for i in range(n_batches):
            k, n = i % 2, grids[i].shape[0]
            if i + 1 < n_batches:
                generate(i + 1)
            main_s.wait_event(ready[k])
            x = bufs[k][:n]
            emb = wb.encode(x)
            sims.append(torch.nn.functional.cosine_similarity(emb, ref.expand_as(emb)))   # the score RISE accumulates
            if i in picks:
                kept[i] = (x[picks[i]].clone(), emb[picks[i]].clone())
            free[k].record(main_s)
        sims = torch.cat(sims)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    out = {'workload': 'RISE-style embeddings sweep, ResNet-101 224x224, synthetic', 'masks': args.masks, 'batch': args.batch,
           'seconds': dt, 'images_per_s': args.masks / dt, 'forward_TFLOP_per_s': args.masks * 14.419e9 / dt / 1e12,
           'frac_of_fp32_mfma_peak': args.masks * 14.419e9 / dt / 157.3e12, 'mean_similarity': float(sims.mean()),
           'forward_split': not args.no_split, 'sweeps_images_per_s': [args.masks / t for t in times],
           'reported': 'median of %d sweeps in one process' % len(times)}
    if kept:
        from oracle import ebp_oracle as O            # the checker, never the thing measured
        ow = O.OracleWhitebox('stresnet101', sd, ('hooked', None), 'affineonly_with_prior')
        worst = 0.0
        for i, (xs, es) in kept.items():
            for r in range(xs.shape[0]):
                want = ow.encode(xs[r:r + 1].cpu()).reshape(-1).numpy()
                got = es[r].reshape(-1).cpu().numpy()
                worst = max(worst, float(np.abs(got - want).max() / np.abs(want).max()))
        out['parity'] = {'probes_checked': sum(len(v) for v in picks.values()), 'max_rel_err_vs_cpu_oracle': worst, 'tolerance': 1e-4}
        assert worst <= 1e-4, worst
    print(json.dumps(out))


if __name__ == '__main__':
    main()
