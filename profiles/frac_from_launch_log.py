#!/usr/bin/env python
"""Recompute bench.py's roofline.frac from the raw launch log of the TIMED schedule (bench.py --launch-log-out FILE: one line per GEMM launch,
start / end in 10 ns ticks of s_memrealtime, written by the kernels while the step runs on its streams exactly as timed).

  python profiles/frac_from_launch_log.py profiles/r5/launch_log.csv [--steps 6] [--alg-gflop 2768.4]

frac = FLOPs per step / (union of the launches' busy intervals per step) / 157.3 TFLOP/s, FLOPs per step = min(algorithmic, executed), executed =
sum over the launches of 2 K M Cout halves (K as packed: the 7x7 stems count 196 rows for 147).  The window is `steps` whole step periods: from
the end of the first recorded step to the end of step `steps` (the log holds steps + 2).  No repo code is imported.
Launches of configuration 9 ran as bf16x6 on the bf16 MFMA pipe (ceiling 2516.6 / 6 = 419.4 fp32-equivalent TFLOP/s): with a share s of the FLOPs on it
the roofline is the harmonic mix 1 / (s / 419.4 + (1 - s) / 157.3); the fraction against the fp32 MFMA peak alone is printed beside it.
The figure rocprofv3 can reproduce is the ONE-stream one (frac_serial: sum of the conv_gemm* durations of kernel_stats_serial.csv)."""
import argparse

ap = argparse.ArgumentParser()
ap.add_argument('csv')
ap.add_argument('--steps', type=int, default=6)
ap.add_argument('--alg-gflop', type=float, default=None, help='algorithmic GFLOP per step (6 F_fwd x batch); default: the executed count')
a = ap.parse_args()
rows = [l.strip().split(',') for l in open(a.csv)][1:]
per_step = len(rows) // (a.steps + 2)
recs = [(int(r[0]), int(r[9]), int(r[10]), 2.0 * int(r[4]) * int(r[5]) * int(r[2]) * int(r[3])) for r in rows if int(r[9]) > 0 and int(r[10]) > 0]
end_of = lambda j: max(e for q, s, e, f in recs if j * per_step <= q < (j + 1) * per_step)      # noqa: E731
t0, t1 = end_of(0), end_of(a.steps)
ev = sorted([(max(s, t0), 1) for q, s, e, f in recs if min(e, t1) > max(s, t0)] + [(min(e, t1), -1) for q, s, e, f in recs if min(e, t1) > max(s, t0)])
busy, depth, last = 0, 0, t0
for t, d in ev:
    busy += (t - last) if depth > 0 else 0
    depth, last = depth + d, t
executed = sum(f for q, s, e, f in recs if per_step <= q < 2 * per_step)
flop = min(executed, a.alg_gflop * 1e9) if a.alg_gflop else executed
print('launches per step %d, step %.3f ms, GEMM busy (union) %.3f ms per step, executed %.1f GFLOP per step, used %.1f' %
      (per_step, (t1 - t0) * 1e-5 / a.steps, busy * 1e-5 / a.steps, executed / 1e9, flop / 1e9))
fl_all = sum(2.0 * int(r[4]) * int(r[5]) * int(r[2]) * int(r[3]) for r in rows)
share = sum(2.0 * int(r[4]) * int(r[5]) * int(r[2]) * int(r[3]) for r in rows if int(r[8]) == 9) / max(fl_all, 1.0)
peak = 1.0 / (share / 419.43e12 + (1.0 - share) / 157.3e12)
ach = flop / (busy * 1e-8 / a.steps)
print('bf16x6 share of the FLOPs %.3f -> roofline %.1f TFLOP/s (fp32 MFMA 157.3, bf16x6 419.4)' % (share, peak / 1e12))
print('achieved %.1f TFLOP/s over the union -> frac %.4f (against the fp32 MFMA peak alone: %.4f)   (over the whole step: %.1f TFLOP/s -> %.4f)' %
      (ach / 1e12, ach / peak, ach / 157.3e12, flop / ((t1 - t0) * 1e-8 / a.steps) / 1e12, flop / ((t1 - t0) * 1e-8 / a.steps) / peak))
