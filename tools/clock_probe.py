#!/usr/bin/env python
"""clock_probe.py -- the shader clock the chip actually sustains while bench.py's workload runs.

Every conv_gemm workgroup records its life in two time bases (xfr_debug_conv_stamps: s_memrealtime = 100 MHz reference clock,
s_memtime = shader-clock cycles); cycles / time = the effective clock of that workgroup's XCD during its life.  The fp32 MFMA
peak is 64 FLOP/clk/SIMD x 1024 SIMDs x clock: 157.3 TFLOP/s at the nominal 2.4 GHz, proportionally less at the clock the
power limit allows.  Runs the ResNet-101 triplet step like bench.py (same engine, batch, streams), `--steps` times with the
stamps on, and prints the distribution over all workgroup records left in the buffer (the last launch that used each index).

    python tools/clock_probe.py [--steps 20] [--serial] [--model resnet101]
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
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--serial', action='store_true', help='one stream (the schedule rocprofv3 sees) instead of the timed three-stream schedule')
    args = ap.parse_args()
    import numpy as np
    import torch
    from xfr_amd import _lib, synth
    from xfr_amd.engine import Engine
    from xfr_amd.models import resnet
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    B = args.batch
    bb = resnet.ResNet([3, 4, 23, 3], num_classes=2)
    prog = bb.build_program()
    eng = Engine(prog, 2 * B, dev)
    eng.load_weights(synth.synth_state_dict(bb, seed=0, recipe='mild'))
    eng.set_mode('affineonly_with_prior')
    if not args.serial:
        eng.set_pipeline(True)
    imgs = synth.bench_images(B, (3, 224, 224), seed=1234, mean=resnet.MEAN_RGB).to(dev)
    gallery, probes = imgs[:2 * B].contiguous(), imgs[2 * B:3 * B].contiguous()
    enc_t = prog.marks['encode']
    if args.serial:
        eng.set_profile(True)

    def step():
        return eng.triplet_contrastive(probes, gallery, enc_t, 1.0 / 2500.0, None, inputs_ready=not args.serial)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    from xfr_amd import tuning
    t0 = time.perf_counter()
    clk = tuning.shader_clock(step, args.steps, dev)
    dt = time.perf_counter() - t0
    out = {'schedule': 'serial (one stream)' if args.serial else 'timed (three streams, pipelined)', 'steps': args.steps,
           'ms_per_step_with_stamps': 1e3 * dt / args.steps, 'shader_clock_GHz': clk,
           'fp32_mfma_peak_at_median_clock_TFLOPs': 64 * 1024 * clk['p50'] * 1e9 / 1e12, 'nominal_peak_TFLOPs': 157.3}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
