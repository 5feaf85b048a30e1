#!/usr/bin/env python
"""Per-shape table of the engine's GEMM launches from an XFR_PROFILE_DUMP csv (one row per launch:
Cout, nhalves, K, M, kh, stride, out_stride, relu_in, accumulate, ms, TFLOP/s)."""
import collections
import sys


def main(path):
    rows = [l.strip().split(',') for l in open(path) if l.strip()]
    agg = collections.OrderedDict()
    for r in rows:
        a = agg.setdefault(tuple(r[:9]), [0, 0.0])
        a[0] += 1
        a[1] += float(r[9])
    tot = sum(a[1] for a in agg.values())
    print('Cout,nhalves,K,M,kh,stride,out_stride,relu_in,accumulate | launches total_ms avg_ms TFLOP/s share')
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fl = 2.0 * int(k[2]) * int(k[3]) * int(k[0]) * int(k[1])
        print(','.join(k), '|', n, '%.3f' % ms, '%.4f' % (ms / n), '%.1f' % (fl / (ms / n * 1e-3) / 1e12), '%.1f%%' % (100 * ms / tot))
    print('total ms', '%.3f' % tot)


if __name__ == '__main__':
    main(sys.argv[1])
