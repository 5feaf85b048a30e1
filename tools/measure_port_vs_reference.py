#!/usr/bin/env python
"""how fast is the CPU oracle (bench.py's cpu_baseline, kind "port") relative to the REAL reference
on identical hardware?  Build container only (needs /root/reference): the same ResNet-101 triplets (2 encodes +
contrastive_ebp, batch 1, mode affineonly_with_prior) through both, same thread count, interleaved.  Writes
profiles/rN/port_vs_reference.json, which bench.py attaches to its cpu_baseline object.

    python tools/measure_port_vs_reference.py [threads] [triplets] [out.json]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, 'profiles', 'r2', 'port_vs_reference.json')
    import torch
    import ref_import
    from make_golden import ref_net
    from oracle import ebp_oracle as O
    from parity_utils import make_backbone
    from xfr_amd import synth
    from xfr_amd.models import resnet
    torch.set_num_threads(threads)
    mode = 'affineonly_with_prior'
    bb, sd = make_backbone('stresnet101', seed=0, num_classes=2)
    imgs = synth.synth_smooth_images(3 * n, (3, 224, 224), seed=1234, mean=resnet.MEAN_RGB)
    mates, nonmates, probes = imgs[:n], imgs[n:2 * n], imgs[2 * n:]
    ns = ref_import.load()
    wbn = ref_net('stresnet101', sd, 2)
    wb = ns.whitebox.Whitebox(wbn, ebp_subtree_mode=mode)
    ow = O.OracleWhitebox('stresnet101', sd, ('hooked', None), mode)

    def ref_triplet(i):
        xm = wbn.encode(mates[i:i + 1]).detach()
        xn = wbn.encode(nonmates[i:i + 1]).detach()
        wbn.set_triplet_classifier((1.0 / 2500.0) * xm, (1.0 / 2500.0) * xn)
        r = wb.contrastive_ebp(probes[i:i + 1], 0, 1)
        wb._ebp_mode = 'disable'
        return r

    def port_triplet(i):
        xm = ow.encode(mates[i:i + 1]) / 2500.0
        xn = ow.encode(nonmates[i:i + 1]) / 2500.0
        ow.set_triplet_classifier(xm, xn)
        return ow.contrastive_ebp(probes[i:i + 1], 0, 1)

    ref_triplet(0), port_triplet(0)          # warm-up
    t_ref = t_port = 0.0
    import numpy as np
    worst = 0.0
    for i in range(n):
        t = time.time(); a = ref_triplet(i); t_ref += time.time() - t
        t = time.time(); b = port_triplet(i); t_port += time.time() - t
        worst = max(worst, float(np.abs(a - b).max() / a.max()))
    res = {'what': 'ResNet-101 triplet (2 encodes + contrastive_ebp), batch 1, %s, %d triplets, interleaved' % (mode, n),
           'threads': threads, 'host': 'build container (%d logical CPUs)' % (os.cpu_count() or 0),
           'reference_maps_s': n / t_ref, 'port_maps_s': n / t_port, 'port_vs_reference': (n / t_port) / (n / t_ref),
           'max_rel_map_difference': worst}
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
