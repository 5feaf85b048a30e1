#!/usr/bin/env python
"""sp2_prof.py -- where a wave of the bf16x6 v2 kernel (conv_gemm_split.hip) spends an interval: a library built with -DSP2_PROF
(tools/build_sp2_variants.sh) accumulates shader cycles per phase; this prints them per K-step pair, per group.

    python tools/sp2_prof.py --lib xfr_amd/csrc/variants/libxfr_amd_PROF.so [--shape 256,14,14,256,3,1,1] [--nb 64]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', required=True)
    ap.add_argument('--shape', default='256,14,14,256,3,1,1')
    ap.add_argument('--nb', type=int, default=64)
    ap.add_argument('--reps', type=int, default=40)
    args = ap.parse_args()
    import numpy as np
    import torch
    from xfr_amd import _lib
    _lib.LIB_PATH = os.path.abspath(args.lib)
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    cin, h, w, cout, k, stride, pad = [int(v) for v in args.shape.split(',')]
    nb = args.nb
    g = torch.Generator().manual_seed(0)
    x = torch.randn((cin, nb, h, w), generator=g).to(dev)
    wt = (torch.randn((cout, cin, k, k), generator=g) / (cin * k * k) ** 0.5).contiguous()
    b = torch.randn((cout,), generator=g)
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    flop = 2.0 * cin * k * k * cout * nb * oh * ow
    out = torch.zeros((cout, nb, oh, ow), device=dev)
    ms = ctypes.c_float()
    _lib.check(lib.xfr_debug_conv(x.data_ptr(), wt.data_ptr(), b.data_ptr(), out.data_ptr(), cin, h, w, nb, cout, k, k, stride, pad, 0, 9, args.reps, ctypes.byref(ms)))
    torch.cuda.synchronize()
    t_ms = ms.value
    nwg = 4096
    st = torch.zeros((4 * nwg * 32,), dtype=torch.int64, device=dev)
    _lib.check(lib.xfr_debug_conv_stamps(st.data_ptr(), nwg))
    _lib.check(lib.xfr_debug_conv(x.data_ptr(), wt.data_ptr(), b.data_ptr(), out.data_ptr(), cin, h, w, nb, cout, k, k, stride, pad, 0, 9, 1, ctypes.byref(ms)))
    torch.cuda.synchronize()
    _lib.check(lib.xfr_debug_conv_stamps(None, 0))
    v = st.cpu().numpy().reshape(4, nwg, 4, 8)
    line = '%-40s %s %.4f ms %6.1f TF-eq' % (os.path.basename(args.lib), args.shape, t_ms, flop / (t_ms * 1e-3) / 1e12)
    for grp in (0, 1):
        r = v[grp]
        ok = r[:, 0, 5] > 0
        if not ok.any():
            continue
        trips = r[ok][:, :, 5].astype(np.float64)
        per = [np.median(r[ok][:, :, i].astype(np.float64) / trips) for i in (1, 2, 3, 4)]
        line += ' | grp %d (%d trips): load issue %5.0f wait+bar %5.0f | compute issue %5.0f wait+bar %5.0f = %5.0f cyc per 2 steps' % (
            grp, int(np.median(trips)), per[0], per[1], per[2], per[3], sum(per))
        d = v[2 + grp][ok]
        det = [np.median(d[:, :, i].astype(np.float64) / trips) for i in range(7)]
        line += ' [load phase: wait X %4.0f, fold %4.0f, frag reads %4.0f, split+store %4.0f, W DMA %4.0f, X loads %4.0f, to end %4.0f]' % tuple(det)
    print(line, flush=True)


if __name__ == '__main__':
    main()
