#!/usr/bin/env python
"""conv_sweep.py -- the implicit-GEMM kernels of xfr_amd/csrc/conv_gemm.hip in isolation, on the GEMM shapes of a
ResNet-101 / ResNet-50-128d / Light-CNN step, through the C ABI's xfr_debug_conv.

For every shape and every tile configuration: `reps` back-to-back launches on the null stream (HIP events), TFLOP/s, and the
largest difference to the first configuration's output (the kernels differ in K summation order only).

    python tools/conv_sweep.py --cfgs 4,6,7 --reps 40 [--set r101|r50|lcnn|all] [--nb 64]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# cin, h, w, cout, k, stride, pad   (forward geometry; a backward-data GEMM of a stride-1 layer has cin/cout swapped)
R101 = [
    (256, 14, 14, 256, 3, 1, 1), (1024, 14, 14, 256, 1, 1, 0), (256, 14, 14, 1024, 1, 1, 0),
    (128, 28, 28, 128, 3, 1, 1), (512, 28, 28, 128, 1, 1, 0), (128, 28, 28, 512, 1, 1, 0),
    (64, 56, 56, 64, 3, 1, 1), (256, 56, 56, 64, 1, 1, 0), (64, 56, 56, 256, 1, 1, 0), (64, 56, 56, 64, 1, 1, 0),
    (512, 7, 7, 512, 3, 1, 1), (2048, 7, 7, 512, 1, 1, 0), (512, 7, 7, 2048, 1, 1, 0),
    (3, 224, 224, 64, 7, 2, 3),
    (256, 56, 56, 128, 1, 2, 0), (512, 28, 28, 256, 1, 2, 0), (1024, 14, 14, 512, 1, 2, 0),
]
LCNN = [
    (128, 16, 16, 256, 3, 1, 1), (192, 16, 16, 384, 3, 1, 1), (96, 32, 32, 192, 3, 1, 1), (48, 64, 64, 96, 3, 1, 1),
    (1, 128, 128, 96, 5, 1, 2), (48, 64, 64, 96, 1, 1, 0), (96, 32, 32, 192, 1, 1, 0), (192, 16, 16, 384, 1, 1, 0),
    (128, 16, 16, 256, 1, 1, 0), (192, 16, 16, 256, 3, 1, 1), (96, 32, 32, 384, 3, 1, 1),
    # backward-data GEMMs onto 96 channels (row 11 ff.): Cout = 96 fills a 64-row tile pair to 75 %
    (192, 32, 32, 96, 3, 1, 1), (384, 32, 32, 96, 3, 1, 1), (192, 64, 64, 96, 1, 1, 0),
]


def stamp_report(lib, _lib, torch, dev, a, cfg, ms_ref):
    """One stamped launch (xfr_debug_conv_stamps): phase durations per wave and the launch's timeline per CU."""
    import numpy as np
    x, wt, b, out, cin, h, w, nbb, cout, k, stride, pad = a
    nwg = 16384
    st = torch.zeros((nwg * 32,), dtype=torch.int64, device=dev)
    _lib.check(lib.xfr_debug_conv_stamps(st.data_ptr(), nwg))
    ms = ctypes.c_float()
    _lib.check(lib.xfr_debug_conv(x.data_ptr(), wt.data_ptr(), b.data_ptr(), out.data_ptr(), cin, h, w, nbb, cout, k, k, stride, pad, 0,
                                  cfg, 1, ctypes.byref(ms)))
    torch.cuda.synchronize()
    _lib.check(lib.xfr_debug_conv_stamps(None, 0))
    v = st.cpu().numpy().reshape(nwg, 4, 8)
    used = v[:, 0, 0] != 0
    bidx = np.nonzero(used)[0]
    v = v[used]
    n = v.shape[0]
    xcc = v[:, 0, 6] & 15
    t = v[:, :, :5].astype(np.float64)
    base = t[:, :, 0].min()
    t = np.where(t > 0, t - base, 0.0)
    t0 = 0.0
    done = t[:, :, 4].max(axis=1) > 0          # non-last tail parts leave before the epilogue
    span = max(t[:, :, 4].max(), t[:, :, 2].max()) - t0
    tick_us = 0.01                             # s_memrealtime: the 100 MHz reference clock, one time base for all XCDs
    f = lambda a_: '%.1f/%.1f/%.1f' % tuple(np.percentile(a_ * tick_us, [10, 50, 90]))
    w0 = t[:, 0, :]
    cu = xcc * 256 + ((v[:, 0, 5] >> 8) & 255)
    ncu = len(set(cu.tolist()))
    lines = ['    cfg %d: %d workgroups on %d CUs, launch %.1f us (avg of reps %.1f), stamped span %.1f us' % (cfg, n, ncu, ms.value * 1e3, ms_ref * 1e3, span * tick_us)]
    lines.append('      start skew p10/50/90 %s us | prologue %s | K loop (slowest wave) %s | exchange %s | epilogue %s' % (
        f(w0[:, 0] - t0), f((t[:, :, 1] - t[:, :, 0]).max(axis=1)), f((t[:, :, 2] - t[:, :, 1]).max(axis=1)),
        f((t[:, :, 3] - t[:, :, 2]).max(axis=1)), f((t[done][:, :, 4] - t[done][:, :, 3]).max(axis=1))))
    ends = (np.maximum(t[:, :, 4].max(axis=1), t[:, :, 2].max(axis=1)) - t0) * tick_us
    lines.append('      workgroup end times p10/50/90/100 %.1f/%.1f/%.1f/%.1f us; K-loop wave skew inside a workgroup p50/p90 %.1f/%.1f us' % (
        *np.percentile(ends, [10, 50, 90, 100]), *np.percentile((t[:, :, 2].max(axis=1) - t[:, :, 2].min(axis=1)) * tick_us, [50, 90])))
    # per CU: when did its last workgroup end, how many workgroups did it run
    cuid = xcc * 256 + ((v[:, 0, 5] >> 8) & 255)         # HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    per = {}
    for i, c_ in enumerate(cuid.tolist()):
        per.setdefault(c_, []).append(i)
    cu_end = np.array([ends[ix].max() for ix in per.values()])
    cu_n = np.array([len(ix) for ix in per.values()])
    lines.append('      per CU: workgroups min/median/max %d/%d/%d; last end p10/50/90/100 %.1f/%.1f/%.1f/%.1f us' % (
        cu_n.min(), np.median(cu_n), cu_n.max(), *np.percentile(cu_end, [10, 50, 90, 100])))
    late = np.argsort(-ends)[:6]
    for i in late:
        lines.append('      straggler block %d: start %.1f prologue %.1f K loop %.1f exchange %.1f epilogue %.1f end %.1f us' % (
            bidx[i], t[i, 0, 0] * tick_us, (t[i, :, 1] - t[i, :, 0]).max() * tick_us, (t[i, :, 2] - t[i, :, 1]).max() * tick_us,
            (t[i, :, 3] - t[i, :, 2]).max() * tick_us, (t[i, :, 4] - t[i, :, 3]).max() * tick_us, ends[i]))
    life = t[:, 0, 4] - t[:, 0, 0]
    okl = (t[:, 0, 4] > 0) & (life > 500)              # lives of at least 5 us: the 10 ns stamp resolution is then < 0.2 %
    if okl.any():
        ghz = v[okl][:, 0, 7].astype(np.float64) / (life[okl] * 10.0)
        lines.append('      effective shader clock over a workgroup\'s life (s_memtime / s_memrealtime): p10/50/90 %.3f/%.3f/%.3f GHz' % tuple(np.percentile(ghz, [10, 50, 90])))
    if os.environ.get('XFR_STAMP_RAW'):
        lines.append('      raw wg0: %s' % v[0].tolist())
        lines.append('      raw wg1: %s' % v[1].tolist())
        lines.append('      t0 %r  max t4 %r max t2 %r' % (t0, t[:, :, 4].max(), t[:, :, 2].max()))
    return '\n'.join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfgs', default='4,6,7')
    ap.add_argument('--reps', type=int, default=40)
    ap.add_argument('--nb', type=int, default=64, help='images per launch (a 32-triplet step: 64 gallery images / 2 x 32 gradient rows)')
    ap.add_argument('--set', default='r101', choices=['r101', 'lcnn', 'all'])
    ap.add_argument('--only', default=None, help='comma list of shape indices')
    ap.add_argument('--passes', type=int, default=2, help='measure the configuration list this many times per shape and report the fastest pass of each (order effects: clocks, caches)')
    ap.add_argument('--stamps', action='store_true', help='per-wave phase stamps of ONE launch per (shape, cfg): where the time goes')
    ap.add_argument('--clock', action='store_true', help='stamps ON during the timed launches: the effective shader clock (s_memtime cycles / s_memrealtime) over the workgroup records left in the buffer')
    args = ap.parse_args()
    import torch
    from xfr_amd import _lib
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    cfgs = [int(c) for c in args.cfgs.split(',')]
    shapes = {'r101': R101, 'lcnn': LCNN, 'all': R101 + LCNN}[args.set]
    if args.only:
        shapes = [shapes[int(i)] for i in args.only.split(',')]
    nb = args.nb
    print('# nb %d reps %d   columns: shape | per cfg: ms TFLOP/s maxdiff-vs-first' % (nb, args.reps))
    tot = {c: 0.0 for c in cfgs}
    for (cin, h, w, cout, k, stride, pad) in shapes:
        nbb = nb * 2 if (cin, h) == (1, 128) else nb
        g = torch.Generator().manual_seed(0)
        x = torch.randn((cin, nbb, h, w), generator=g).to(dev)
        wt = (torch.randn((cout, cin, k, k), generator=g) / (cin * k * k) ** 0.5).contiguous()
        b = torch.randn((cout,), generator=g)
        oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        flop = 2.0 * cin * k * k * cout * nbb * oh * ow
        first = None
        cells = []
        for c in cfgs:
            out = torch.zeros((cout, nbb, oh, ow), device=dev)
            ms = ctypes.c_float()
            if args.clock:
                nwg = 16384
                stc = torch.zeros((nwg * 32,), dtype=torch.int64, device=dev)
                _lib.check(lib.xfr_debug_conv_stamps(stc.data_ptr(), nwg))
            best = None
            for _ in range(max(1, args.passes if not (args.clock or args.stamps) else 1)):
                _lib.check(lib.xfr_debug_conv(x.data_ptr(), wt.data_ptr(), b.data_ptr(), out.data_ptr(), cin, h, w, nbb, cout, k, k, stride,
                                              pad, 0, c, args.reps, ctypes.byref(ms)))
                torch.cuda.synchronize()
                best = ms.value if best is None else min(best, ms.value)
            ms.value = best
            if args.clock:
                import numpy as np
                _lib.check(lib.xfr_debug_conv_stamps(None, 0))
                v = stc.cpu().numpy().reshape(nwg, 4, 8)
                life = (v[:, :, 4] - v[:, :, 0]).astype(np.float64).ravel()
                cyc = v[:, :, 7].astype(np.float64).ravel()
                ok = (life > 1000) & (cyc > 0)
                ghz = cyc[ok] / (life[ok] * 10.0)
                ghz = ghz[(ghz > 0.5) & (ghz < 4.0)]      # concurrent streams overwrite each other's records: mixed ones fall outside
                print('    cfg %d: effective shader clock during the %d timed launches p10/50/90 %.3f/%.3f/%.3f GHz (%d records)' % (
                    (c, args.reps) + tuple(np.percentile(ghz, [10, 50, 90])) + (len(ghz),)))
            if args.stamps:
                print(stamp_report(lib, _lib, torch, dev, (x, wt, b, out, cin, h, w, nbb, cout, k, stride, pad), c, ms.value))
            if first is None:
                first = out
                d = 0.0
            else:
                d = float((out - first).abs().max() / first.abs().max())
            tot[c] += ms.value
            cells.append('%3d: %7.4f ms %6.1f TF %.1e' % (c, ms.value, flop / (ms.value * 1e-3) / 1e12, d))
        print('%-28s K %5d M %7d N %5d | %s' % (str((cin, h, w, cout, k, stride)), cin * k * k, nbb * oh * ow, cout, ' | '.join(cells)), flush=True)
    print('# sum of per-launch ms: ' + '  '.join('%d: %.3f' % (c, tot[c]) for c in cfgs))


if __name__ == '__main__':
    main()
