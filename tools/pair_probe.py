#!/usr/bin/env python
"""pair_probe.py -- do GEMM launches of DIFFERENT layers on different streams help each other?

Two layer shapes, each on two streams of back-to-back launches (xfr_debug_conv, cfg 2000000 + tile configuration), first alone,
then both at once from two host threads.  Round 3 on MI355X: the aggregate of the mix is the average of the parts (123.8 TFLOP/s
for the layer-3 3x3 at 133 and the K = 256 expansion at 118) -- a launch gains from a second launch of its own kind (3x3: 117.7
-> 133, K = 256: 100 -> 118) and nothing more from a different one.

    python tools/pair_probe.py
"""
import ctypes, sys, threading, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xfr_amd import _lib
lib = _lib.load()
dev = torch.device('cuda', 0)
def mk(cin, h, w, cout, k, stride, pad, nb=64):
    g = torch.Generator().manual_seed(0)
    x = torch.randn((cin, nb, h, w), generator=g).to(dev)
    wt = (torch.randn((cout, cin, k, k), generator=g) / (cin * k * k) ** 0.5).contiguous()
    b = torch.randn((cout,), generator=g)
    oh = (h + 2 * pad - k) // stride + 1
    out = torch.zeros((cout, nb, oh, oh), device=dev)
    fl = 2.0 * cin * k * k * cout * nb * oh * oh
    return (x, wt, b, out, cin, h, w, nb, cout, k, stride, pad, fl)
def run(a, cfg, reps, res, key):
    x, wt, b, out, cin, h, w, nb, cout, k, stride, pad, fl = a
    ms = ctypes.c_float()
    t0 = time.perf_counter()
    _lib.check(lib.xfr_debug_conv(x.data_ptr(), wt.data_ptr(), b.data_ptr(), out.data_ptr(), cin, h, w, nb, cout, k, k, stride, pad, 0, cfg, reps, ctypes.byref(ms)))
    res[key] = (ms.value, time.perf_counter() - t0)
A = mk(256, 14, 14, 256, 3, 1, 1)      # K 2304, 784 tiles (split-K kernel)
Bs = mk(256, 14, 14, 1024, 1, 1, 0)    # K 256, 3136 tiles (K1)
C = mk(1024, 14, 14, 256, 1, 1, 0)     # K 1024, 784 tiles (K1)
R = 4000
for name, (p, cp), (q, cq) in (('3x3 K2304 (ks) + 1x1 K256 (k1)', (A, 2000007), (Bs, 2000004)), ('3x3 K2304 (ks) + 1x1 K1024 (k1)', (A, 2000007), (C, 2000004)),
                               ('1x1 K256 + 1x1 K1024', (Bs, 2000004), (C, 2000004))):
    res = {}
    run(p, cp, 200, res, 'w'); run(q, cq, 200, res, 'w')
    run(p, cp, R, res, 'p'); run(q, cq, R, res, 'q')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(p, cp, R, res, 'pp')), threading.Thread(target=run, args=(q, cq, R, res, 'qq'))]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    both = time.perf_counter() - t0
    tp, tq = res['p'][0] * 2 * R * 1e-3, res['q'][0] * 2 * R * 1e-3        # ms is per launch, two streams of R launches each
    print('%s: alone (2 streams each) %.3f + %.3f = %.3f s (%.1f / %.1f TF); together %.3f s = %.1f TF aggregate; per-launch ms alone %.4f %.4f together %.4f %.4f' % (
        name, tp, tq, tp + tq, p[12] / (res['p'][0] * 1e-3) / 1e12, q[12] / (res['q'][0] * 1e-3) / 1e12, both, (p[12] + q[12]) * 2 * R / both / 1e12,
        res['p'][0], res['q'][0], res['pp'][0], res['qq'][0]))
