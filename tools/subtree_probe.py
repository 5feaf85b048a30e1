#!/usr/bin/env python
"""Where the time of weighted_subtree_ebp_batch goes (BASELINE.json configs[4]'s dominant method): N probes on ResNet-101 ('norelu', top-32), wall
time per phase (device synchronised between phases) and, from the kernels' own launch log, the GEMM launches by shape / chain / kernel.
    python tools/subtree_probe.py [--n 8] [--topk 32] [--reps 3] [--log]"""
import argparse
import collections
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=8)
    ap.add_argument('--topk', type=int, default=32)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--max-batch', type=int, default=128)
    ap.add_argument('--sweep-batch', type=int, default=None, help='candidate layers per probe in the first layerwise round (default: min(2 * max_batch / n, 2 * topk))')
    ap.add_argument('--log', action='store_true', help='per-shape table of the GEMM launches of one call (in-kernel launch log)')
    args = ap.parse_args()
    import numpy as np
    import torch
    from xfr_amd import synth, tuning
    from xfr_amd.engine import Engine
    from xfr_amd.models import resnet, whitebox as WB
    dev = torch.device('cuda', 0)
    bb = resnet.ResNet([3, 4, 23, 3], num_classes=2)
    bb.load_state_dict(synth.synth_state_dict(bb, seed=0))
    bb.to(dev)
    wbn = WB.WhiteboxSTResnet(bb)
    wbn.default_max_batch = args.max_batch
    wb = WB.Whitebox(wbn, ebp_subtree_mode='norelu')
    n = args.n
    x = synth.synth_smooth_images(n, (3, 224, 224), seed=3, mean=resnet.MEAN_RGB).to(dev)
    xm, xn = synth.unit_rows(n, 512, seed=1).to(dev), synth.unit_rows(n, 512, seed=2).to(dev)
    eng = wb._engine(n)
    # instrument the engine's phases
    T = collections.OrderedDict()
    for name in ('subtree_weights', 'ebp_capture', 'layerwise'):
        fn = getattr(eng, name)

        def timed(*a, _fn=fn, _name=name, **k):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = _fn(*a, **k)
            torch.cuda.synchronize()
            T.setdefault(_name, [0.0, 0])
            T[_name][0] += time.perf_counter() - t
            T[_name][1] += 1
            return r
        setattr(eng, name, timed)
    wb.weighted_subtree_ebp_batch(x, xm, xn, topk=args.topk, sweep_batch=args.sweep_batch)          # warm
    T.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        res = wb.weighted_subtree_ebp_batch(x, xm, xn, topk=args.topk, sweep_batch=args.sweep_batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    out = {'probes': n, 'topk': args.topk, 'ms_per_call': 1e3 * dt, 'ms_per_probe': 1e3 * dt / n,
           'phases_ms_per_call': {k: round(1e3 * v[0] / args.reps, 2) for k, v in T.items()}, 'phase_calls_per_call': {k: v[1] / args.reps for k, v in T.items()},
           'valid_subtrees': [len(r[3]) for r in res]}
    out['phases_ms_per_call']['host + merge (rest)'] = round(1e3 * dt - sum(out['phases_ms_per_call'].values()), 2)
    print(json.dumps(out))
    if args.log:
        csv = tuning.record_launch_log(lambda: wb.weighted_subtree_ebp_batch(x, xm, xn, topk=args.topk, sweep_batch=args.sweep_batch), 0, dev, launches_per_step_cap=40000)
        agg = collections.OrderedDict()
        for l in list(open(csv))[1:]:
            r = l.strip().split(',')
            a, b = int(r[9]), int(r[10])
            if a <= 0 or b <= a:
                continue
            key = (r[2], r[3], r[4], r[5], r[6], 'chain' + r[7], 'cfg' + r[8])
            e = agg.setdefault(key, [0, 0.0, 0.0])
            e[0] += 1
            e[1] += (b - a) * 1e-5
            e[2] += 2.0 * int(r[4]) * int(r[5]) * int(r[2]) * int(r[3])
        os.remove(csv)
        tot = sum(v[1] for v in agg.values())
        fl = sum(v[2] for v in agg.values())
        print('GEMM launches of 2 calls: %d, %.2f ms, %.1f GFLOP executed (%.1f TFLOP/s while a GEMM runs)' % (sum(v[0] for v in agg.values()), tot, fl / 1e9, fl / tot / 1e9))
        print('Cout,halves,K,M,kh,chain,cfg | launches total_ms avg_ms TFLOP/s share')
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            print(','.join(k), '|', v[0], '%.3f' % v[1], '%.4f' % (v[1] / v[0]), '%.1f' % (v[2] / v[1] / 1e9), '%.1f%%' % (100 * v[1] / tot))


if __name__ == '__main__':
    main()
