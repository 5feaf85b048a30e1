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
    from xfr_amd import tuning
    tuning.record_launch_log(step, steps, dev, csv_path)
    return csv_path


def analyse(csv_path, steps, flop_per_step):
    from xfr_amd import tuning
    return tuning.analyse_launch_log(csv_path, steps, flop_per_step)


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
