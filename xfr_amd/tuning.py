"""Measurement helpers on top of the library's tuning hooks (include/xfr_amd.h: xfr_debug_conv_log, xfr_debug_conv_stamps).

rocprofv3 serialises the queues it traces, and HIP events on one stream say nothing about what the other streams did meanwhile.
The figures bench.py reports for the TIMED schedule therefore come from inside the kernels:

* launch log -- block 0 of every GEMM launch notes when it started, a sample of its workgroups when they ended
  (s_memrealtime: one 100 MHz time base for the whole chip).  From it: the union of the GEMM busy intervals per step (the time
  during which at least one GEMM launch was running; never more than the step itself), how many launches overlapped for how long,
  and per stream the launch time and the gaps.
* shader clock -- every workgroup also counts its life in shader-clock cycles (s_memtime); cycles / time is the clock the chip
  really ran at, which is what the fp32 MFMA peak (64 FLOP/clk/SIMD x 1024 SIMDs x clock) scales with.
"""
import collections
import ctypes
import os
import tempfile

import numpy as np
import torch

from . import _lib


def record_launch_log(step, steps, device, csv_path=None, launches_per_step_cap=1200):
    """Run `step()` steps + 2 times with the launch log on; returns the CSV path (one line per GEMM launch, enqueue order)."""
    lib = _lib.load()
    cap = launches_per_step_cap * (steps + 2)
    log = torch.zeros((cap * 8,), dtype=torch.int64, device=device)
    torch.cuda.synchronize(device)
    _lib.check(lib.xfr_debug_conv_log(log.data_ptr(), cap, None))
    try:
        for _ in range(steps + 2):
            step()
        torch.cuda.synchronize(device)
        if csv_path is None:
            fd, csv_path = tempfile.mkstemp(suffix='.csv', prefix='xfr_gemm_log_')
            os.close(fd)
        _lib.check(lib.xfr_debug_conv_log(None, 0, csv_path.encode()))
    finally:
        lib.xfr_debug_conv_log(None, 0, None)
    return csv_path


def analyse_launch_log(csv_path, steps, flop_per_step):
    """Steady-state figures over `steps` step periods of a log recorded with record_launch_log(steps)."""
    rows = [l.strip().split(',') for l in open(csv_path)][1:]
    # seq, stream, Cout, nhalves, K, M, kh, chain, cfg, start, end  (10 ns ticks)
    recs = [(int(r[0]), r[1], int(r[2]), int(r[3]), int(r[4]), int(r[5]), int(r[9]), int(r[10])) for r in rows if int(r[9]) > 0 and int(r[10]) > 0]
    # share of the GEMM FLOPs (2 Cout nhalves K M per launch) that ran on the bf16x6 kernel (configuration 9, conv_gemm_split.hip K17)
    fl = [(2.0 * int(r[2]) * int(r[3]) * int(r[4]) * int(r[5]), int(r[8])) for r in rows]
    split_share = sum(f for f, c in fl if c == 9) / max(sum(f for f, _ in fl), 1.0)
    split_launches = sum(1 for _, c in fl if c == 9) / float(steps + 2)
    per_step = len(rows) // (steps + 2)
    # window: from the end of recorded step 0 to the end of recorded step `steps` -- exactly `steps` step periods (the forwards of
    # a step run under the previous step's sweep, so a step's own launches span about two periods)
    def end_of(j):
        return max(r[7] for r in recs if j * per_step <= r[0] < (j + 1) * per_step)
    t0, t1 = end_of(0), end_of(steps)
    ev = []
    for r in recs:
        a, b = max(r[6], t0), min(r[7], t1)
        if b > a:
            ev.append((a - t0, 1))
            ev.append((b - t0, -1))
    ev.sort()
    depth, last, hist = 0, 0, collections.Counter()
    for t, d in ev:
        hist[depth] += t - last
        last = t
        depth += d
    span = t1 - t0
    hist[0] += span - last
    busy = sum(v for k, v in hist.items() if k > 0)
    inside = [r for r in recs if r[7] > t0 and r[6] < t1]
    n_in = sum(1 for r in recs if t0 < r[7] <= t1)
    out = {'steps': steps, 'launches_per_step': per_step, 'ms_per_step': span * 1e-5 / steps,
           'gemm_union_busy_ms_per_step': busy * 1e-5 / steps, 'gemm_union_busy_frac': busy / span,
           'avg_launch_ms_in_union': busy * 1e-5 / max(n_in, 1),
           'concurrent_launches_ms_per_step': {str(k): v * 1e-5 / steps for k, v in sorted(hist.items())},
           'achieved_over_union_TFLOPs': flop_per_step / (busy * 1e-8 / steps) / 1e12,
           'achieved_over_step_TFLOPs': flop_per_step / (span * 1e-8 / steps) / 1e12, 'streams': [],
           'split_flop_share': split_share, 'split_launches_per_step': split_launches}
    streams = collections.OrderedDict()
    for r in inside:
        streams.setdefault(r[1], []).append(r)
    for sname, rs in streams.items():
        rs.sort(key=lambda r: r[6])
        dur = sum(min(r[7], t1) - max(r[6], t0) for r in rs)
        gaps = [max(0, b[6] - a[7]) for a, b in zip(rs, rs[1:])] or [0]
        out['streams'].append({'stream': sname, 'dual_launches': sum(1 for r in rs if r[3] == 2) / steps,
                               'launches_per_step': len(rs) / steps, 'sum_launch_ms_per_step': dur * 1e-5 / steps,
                               'sum_gap_ms_per_step': sum(gaps) * 1e-5 / steps, 'median_gap_us': float(np.median(gaps)) * 1e-2,
                               'p90_gap_us': float(np.percentile(gaps, 90)) * 1e-2})
    return out


def shader_clock(step, steps, device, max_workgroups=16384):
    """Effective shader clock (GHz) of the GEMM workgroups while `step()` runs `steps` times: distribution over the workgroup
    records left in the stamp buffer (lives of at least 10 us)."""
    lib = _lib.load()
    st = torch.zeros((max_workgroups * 32,), dtype=torch.int64, device=device)
    torch.cuda.synchronize(device)
    _lib.check(lib.xfr_debug_conv_stamps(st.data_ptr(), max_workgroups))
    try:
        for _ in range(steps):
            step()
        torch.cuda.synchronize(device)
    finally:
        lib.xfr_debug_conv_stamps(None, 0)
    v = st.cpu().numpy().reshape(max_workgroups, 4, 8)
    life = (v[:, 0, 4] - v[:, 0, 0]).astype(np.float64)         # 10 ns ticks
    ok = (v[:, 0, 4] > 0) & (v[:, 0, 0] > 0) & (life > 1000)
    if not ok.any():
        return None
    ghz = v[ok][:, 0, 7].astype(np.float64) / (life[ok] * 10.0)
    p = np.percentile(ghz, [5, 50, 95])
    return {'p5': float(p[0]), 'p50': float(p[1]), 'p95': float(p[2]), 'mean': float(ghz.mean()), 'workgroup_records': int(ok.sum())}
