#!/usr/bin/env python
"""Error of the GEMM kernels against a float64 convolution (CPU): RMS and mean signed error of the fp32 MFMA kernels (cfg 4, 7) and the bf16x6 kernel
(cfg 9) on deep-K layer shapes, with post-ReLU-like (half zero, non-negative) and signed inputs.   python tools/conv_error_probe.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--extra', action='store_true', help='also odd K-step counts, M tails, relu on the input, tiny and large grids (whole tiles and stream-K shares)')
    ap.add_argument('--cfgs', default='4,7,9')
    args = ap.parse_args()
    import numpy as np
    import torch
    from xfr_amd import _lib
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    torch.set_num_threads(16)
    shapes = [(256, 14, 14, 8, 256, 3, 1, 0), (1024, 14, 14, 8, 256, 1, 0, 0), (128, 28, 28, 4, 128, 3, 1, 0)]
    if args.extra:       # (.., relu_in)
        shapes += [(48, 20, 20, 2, 128, 3, 1, 0), (64, 14, 14, 3, 128, 3, 1, 1), (16, 15, 15, 5, 256, 5, 2, 0), (1024, 14, 14, 64, 256, 1, 0, 0),
                   (256, 14, 14, 1, 256, 3, 1, 0), (256, 14, 14, 64, 256, 3, 1, 0), (128, 28, 28, 66, 128, 3, 1, 1), (2048, 14, 14, 2, 512, 1, 0, 0)]
    for (cin, h, w, nb, cout, k, pad, relu_in) in shapes:
        for kind in (('relu', 'signed', 'wide') if not relu_in else ('signed',)):
            g = torch.Generator().manual_seed(1)
            x = torch.randn((nb, cin, h, w), generator=g, dtype=torch.float64)
            if kind == 'relu':
                x = x.clamp_min(0)
            if kind == 'wide':            # gradient-like: magnitudes over many decades
                x = x * torch.exp(8 * torch.randn((nb, cin, h, w), generator=g, dtype=torch.float64))
            x = x.float()
            xin = x.clamp_min(0) if relu_in else x
            wt = (torch.randn((cout, cin, k, k), generator=g) / (cin * k * k) ** 0.5).contiguous()
            want = torch.nn.functional.conv2d(xin.double(), wt.double(), None, padding=pad)
            mag = torch.nn.functional.conv2d(xin.double().abs(), wt.double().abs(), None, padding=pad)       # sum of |terms|
            xg = x.to(dev).permute(1, 0, 2, 3).contiguous()
            line = '%-30s %-6s' % ((cin, h, nb, cout, k, relu_in), kind)
            for cfg in [int(c) for c in args.cfgs.split(',')]:
                out = torch.zeros((cout, nb) + tuple(want.shape[2:]), device=dev)
                ms = ctypes.c_float()
                _lib.check(lib.xfr_debug_conv(xg.data_ptr(), wt.data_ptr(), None, out.data_ptr(), cin, h, w, nb, cout, k, k, 1, pad, relu_in, cfg, 1, ctypes.byref(ms)))
                got = out.permute(1, 0, 2, 3).cpu().double()
                e = (got - want) / mag
                line += ' | cfg %d: rms %.2e mean %+.2e max %.2e' % (cfg, float(e.pow(2).mean().sqrt()), float(e.mean()), float(e.abs().max()))
            print(line, flush=True)


if __name__ == '__main__':
    main()
