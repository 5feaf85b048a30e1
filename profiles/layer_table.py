#!/usr/bin/env python
"""Per-shape table of the engine's GEMM launches, serial (one-stream) schedule.

Two inputs:
* the engine's HIP-event csv (bench.py --serial --profile-csv: Cout,nhalves,K,M,kh,stride,out_stride,relu_in,accumulate,ms,TFLOP/s), one row per
  launch.  HIP events misread the FIRST GEMM of a step: its start event is recorded on a queue that has just been idle, and the launch is charged
  the wake-up (round 3: the stems read 0.87 ms where rocprofv3 saw 0.37; a hipStreamSynchronize in front of it made it 1.03, round 4);
* the in-kernel launch log (bench.py --serial --launch-log-csv: seq,stream,Cout,nhalves,K,M,kh,chain,cfg,start_10ns,end_10ns): block 0's entry to
  the last sampled workgroup's exit, s_memrealtime -- what the kernel itself saw, first launch included.  `--inkernel` selects this format
  (auto-detected from the header).  K is the packed K here (stems: 196 rows for 147), FLOPs are counted with it.
"""
import collections
import sys


def table(agg, title):
    tot = sum(a[1] for a in agg.values())
    print(title)
    for k, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(','.join(k), '|', n, '%.3f' % ms, '%.4f' % (ms / n), '%.1f' % (fl / (ms / n * 1e-3) / 1e12), '%.1f%%' % (100 * ms / tot))
    print('total ms', '%.3f' % tot)


def main(path):
    lines = [l.strip() for l in open(path) if l.strip()]
    agg = collections.OrderedDict()
    if lines and lines[0].startswith('seq,'):
        for l in lines[1:]:
            r = l.split(',')
            a, b = int(r[9]), int(r[10])
            if a <= 0 or b <= a:
                continue
            key = (r[2], r[3], r[4], r[5], r[6], 'chain' + r[7], 'cfg' + r[8])
            e = agg.setdefault(key, [0, 0.0, 0.0])
            e[0] += 1
            e[1] += (b - a) * 1e-5
            e[2] = 2.0 * int(r[4]) * int(r[5]) * int(r[2]) * int(r[3])
        table(agg, 'Cout,nhalves,K(packed),M,kh,chain steps,cfg | launches total_ms avg_ms TFLOP/s share   (in-kernel s_memrealtime stamps)')
        return
    for l in lines:
        r = l.split(',')
        e = agg.setdefault(tuple(r[:9]), [0, 0.0, 0.0])
        e[0] += 1
        e[1] += float(r[9])
        e[2] = 2.0 * int(r[2]) * int(r[3]) * int(r[0]) * int(r[1])
    table(agg, 'Cout,nhalves,K,M,kh,stride,out_stride,relu_in,accumulate | launches total_ms avg_ms TFLOP/s share   (HIP events)')


if __name__ == '__main__':
    main(sys.argv[-1])
