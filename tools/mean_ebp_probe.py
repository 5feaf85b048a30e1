#!/usr/bin/env python
"""Where the time of the job mix's meanEBP goes (EBP of N probes over the 65359-way HOOKED classifier, generate_whitebox_saliency.py:207-214):
wall time per call and the GEMM launches by shape (in-kernel launch log).   python tools/mean_ebp_probe.py [--n 8]"""
import argparse
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=8)
    ap.add_argument('--num-classes', type=int, default=65359)
    args = ap.parse_args()
    import torch
    from xfr_amd import synth, tuning
    from xfr_amd.models import resnet, whitebox as WB
    dev = torch.device('cuda', 0)
    bb = resnet.ResNet([3, 4, 23, 3], num_classes=args.num_classes)
    bb.load_state_dict(synth.synth_state_dict(bb, seed=0))
    bb.to(dev)
    wbn = WB.WhiteboxSTResnet(bb)
    wbn.default_max_batch = 128
    wb = WB.Whitebox(wbn, ebp_subtree_mode='norelu')
    x = synth.synth_smooth_images(args.n, (3, 224, 224), seed=3, mean=resnet.MEAN_RGB).to(dev)
    P = torch.ones((1, args.num_classes))
    call = lambda: wb.ebp(x, P)      # noqa: E731
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    print('meanEBP of %d probes over %d classes: %.2f ms per call, %.2f ms per probe' % (args.n, args.num_classes, 1e3 * (time.perf_counter() - t0) / 5, 1e3 * (time.perf_counter() - t0) / 5 / args.n))
    csv = tuning.record_launch_log(call, 0, dev, launches_per_step_cap=4000)
    agg = collections.OrderedDict()
    for l in list(open(csv))[1:]:
        r = l.strip().split(',')
        a, b = int(r[9]), int(r[10])
        if a <= 0 or b <= a:
            continue
        key = (r[2], r[3], r[4], r[5], r[6], 'chain' + r[7], 'cfg' + r[8])
        e = agg.setdefault(key, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += (b - a) * 1e-5
        e[2] += 2.0 * int(r[4]) * int(r[5]) * int(r[2]) * int(r[3])
    os.remove(csv)
    tot = sum(v[1] for v in agg.values())
    print('GEMM launches of 2 calls: %d, %.2f ms' % (sum(v[0] for v in agg.values()), tot))
    print('Cout,halves,K,M,kh,chain,cfg | launches total_ms avg_ms TFLOP/s share')
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(','.join(k), '|', v[0], '%.3f' % v[1], '%.4f' % (v[1] / v[0]), '%.1f' % (v[2] / v[1] / 1e9), '%.1f%%' % (100 * v[1] / tot))


if __name__ == '__main__':
    main()
