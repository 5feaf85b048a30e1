#!/usr/bin/env python
"""One triplet per call at batch 1 (the reference's API: whitebox.py:506-527): wall time with the device synchronised after every call, and the calling
thread's share of it.  Under `rocprofv3 --kernel-trace --stats --output-format csv` the per-kernel table of exactly these calls.
    python tools/one_triplet_probe.py [--model resnet101|resnet50_128] [--reps 50] [--pipeline 0|1]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='resnet101')
    ap.add_argument('--reps', type=int, default=50)
    ap.add_argument('--pipeline', type=int, default=1)
    ap.add_argument('--triplets', type=int, default=1, help='triplets per call (1 = the reference API)')
    ap.add_argument('--max-batch', type=int, default=8, help='engine size: triplets per call it could take (memory only; the calls are batch 1)')
    a = ap.parse_args()
    import torch
    import bench
    dev = torch.device('cuda', 0)
    args = argparse.Namespace(model=a.model, mode=None, batch=a.max_batch)
    W = bench.make_workload(args, dev, 0)
    W.eng.set_pipeline(a.pipeline)
    fn = W.one
    if a.triplets > 1:
        n, B = a.triplets, W.B
        step_args = W.step_inputs(n) if hasattr(W, 'step_inputs') else None
        if step_args is None:
            raise SystemExit('this workload has no step_inputs')
        fn = step_args
    ref = fn().clone()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts, th = [], []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        s = fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
        th.append(1e3 * (t1 - t0))             # the calling thread's share: enqueueing the call
    ts.sort()
    th.sort()
    print(json.dumps({'model': a.model, 'triplets': a.triplets, 'pipeline': a.pipeline, 'ms_median': ts[len(ts) // 2], 'ms_min': ts[0], 'ms_p90': ts[int(0.9 * (len(ts) - 1))], 'host_ms_median': th[len(th) // 2],
                      'same_bits': bool(torch.equal(s, ref))}))


if __name__ == '__main__':
    main()
