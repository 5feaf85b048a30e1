#!/usr/bin/env python
"""gemm_timeline.py -- what the device did with the GEMM launches of bench.py's timed, three-stream schedule.

rocprofv3 serialises the queues it traces, so its kernel durations describe a schedule that was never timed.  This tool records,
inside the kernels, when the first workgroup of every GEMM launch started and when its last one ended (xfr_debug_conv_log,
s_memrealtime: one 100 MHz time base for the whole chip) while the step runs exactly as bench.py times it, and reports over the
recorded steady-state steps: the union of the GEMM busy intervals (the time during which at least one GEMM launch was running --
`roofline.frac` of the bench line is the algorithmic FLOPs over THIS time), how many launches overlapped for how long, and per
stream the sum of the launch durations and of the gaps between consecutive launches.

    python tools/gemm_timeline.py [--steps 6] [--csv out.csv] [--json]
"""
import argparse
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def record(steps, batch=32, serial=False, csv_path='/tmp/gemm_log.csv', model='resnet101'):
    import torch
    from xfr_amd import _lib, synth
    from xfr_amd.engine import Engine
    from xfr_amd.models import resnet
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    B = batch
    bb = resnet.ResNet([3, 4, 23, 3], num_classes=2)
    prog = bb.build_program()
    eng = Engine(prog, 2 * B, dev)
    eng.load_weights(synth.synth_state_dict(bb, seed=0, recipe='mild'))
    eng.set_mode('affineonly_with_prior')
    if not serial:
        eng.set_pipeline(True)
    imgs = synth.bench_images(B, (3, 224, 224), seed=1234, mean=resnet.MEAN_RGB).to(dev)
    gallery, probes = imgs[:2 * B].contiguous(), imgs[2 * B:3 * B].contiguous()
    enc_t = prog.marks['encode']
    if serial:
        eng.set_profile(True)

    def step():
        return eng.triplet_contrastive(probes, gallery, enc_t, 1.0 / 2500.0, None, inputs_ready=not serial)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    cap = 400 * (steps + 2)
    log = torch.zeros((cap * 2,), dtype=torch.int64, device=dev)
    _lib.check(lib.xfr_debug_conv_log(log.data_ptr(), cap, None))
    for _ in range(steps + 2):
        step()
    torch.cuda.synchronize()
    _lib.check(lib.xfr_debug_conv_log(None, 0, csv_path.encode()))
    return csv_path


def analyse(csv_path, steps, flop_per_step):
    import numpy as np
    rows = [l.strip().split(',') for l in open(csv_path)][1:]
    recs = [(int(r[0]), r[1], int(r[2]), int(r[3]), int(r[4]), int(r[5]), int(r[9]), int(r[10])) for r in rows]
    per_step = len(recs) // (steps + 2)
    recs = recs[per_step:per_step * (steps + 1)]           # drop the first and the last recorded step (pipeline fill / drain)
    t0 = min(r[6] for r in recs)
    ev = []
    for r in recs:
        ev.append((r[6] - t0, 1))
        ev.append((r[7] - t0, -1))
    ev.sort()
    depth, last, hist = 0, 0, collections.Counter()
    for t, d in ev:
        hist[depth] += t - last
        last = t
        depth += d
    span = max(r[7] for r in recs) - t0
    busy = sum(v for k, v in hist.items() if k > 0)
    streams = collections.OrderedDict()
    for r in recs:
        streams.setdefault(r[1], []).append(r)
    out = {'steps': steps, 'launches_per_step': per_step, 'window_ms': span * 1e-5, 'ms_per_step': span * 1e-5 / steps,
           'gemm_union_busy_ms_per_step': busy * 1e-5 / steps, 'gemm_union_busy_frac': busy / span,
           'concurrency_ms_per_step': {str(k): v * 1e-5 / steps for k, v in sorted(hist.items())},
           'achieved_over_union_TFLOPs': flop_per_step / (busy * 1e-8 / steps) / 1e12,
           'achieved_over_window_TFLOPs': flop_per_step / (span * 1e-8 / steps) / 1e12, 'streams': []}
    for sname, rs in streams.items():
        rs.sort(key=lambda r: r[6])
        dur = sum(r[7] - r[6] for r in rs)
        gaps = [max(0, b[6] - a[7]) for a, b in zip(rs, rs[1:])]
        big = sorted(gaps)[-max(1, len(gaps) // 100):]
        out['streams'].append({'stream': sname, 'launches_per_step': len(rs) / steps, 'sum_launch_ms_per_step': dur * 1e-5 / steps,
                               'sum_gap_ms_per_step': sum(gaps) * 1e-5 / steps, 'median_gap_us': float(np.median(gaps)) * 1e-2,
                               'p90_gap_us': float(np.percentile(gaps, 90)) * 1e-2, 'largest_1pct_gaps_ms_per_step': sum(big) * 1e-5 / steps})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--serial', action='store_true')
    ap.add_argument('--csv', default='/tmp/gemm_log.csv')
    args = ap.parse_args()
    record(args.steps, args.batch, args.serial, args.csv)
    print(json.dumps(analyse(args.csv, args.steps, 6 * 14.419e9 * args.batch)))


if __name__ == '__main__':
    main()
