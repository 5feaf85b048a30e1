#!/usr/bin/env python
"""The reference's demo sequence through the Python mirror (demo/test_whitebox.py:124-133): two encode() calls, set_triplet_classifier, contrastive_ebp -- four
calls whose forwards run one after the other on the caller's stream -- with host and with device tensors, device synchronised at the end of each sequence.
The fused entry point (Engine.triplet_contrastive: the three forwards on three streams) is what tools/one_triplet_probe.py times."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import torch
from parity_utils import make_backbone, make_images
from xfr_amd.models import whitebox as WB
dev = 'cuda:0'
bb, sd = make_backbone('stresnet101', seed=0, num_classes=2)
bb.to(dev)
wbn = WB.WhiteboxSTResnet(bb)
wb = WB.Whitebox(wbn, ebp_subtree_mode='affineonly_with_prior')
x = make_images('stresnet101', 3, seed=5)
xm, xn, xp = x[0:1], x[1:2], x[2:3]
def seq(xm, xn, xp):
    em = wbn.encode(xm); en = wbn.encode(xn)
    wbn.set_triplet_classifier((1.0 / 2500.0) * em, (1.0 / 2500.0) * en)
    return wb.contrastive_ebp(xp, k_poschannel=0, k_negchannel=1)
for where, (a, b, c) in (('host tensors', (xm, xn, xp)), ('device tensors', (xm.to(dev), xn.to(dev), xp.to(dev)))):
    for _ in range(5): s = seq(a, b, c)
    torch.cuda.synchronize(); ts = []
    for _ in range(30):
        t0 = time.perf_counter(); s = seq(a, b, c); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    print(json.dumps({'what': 'demo sequence (2 encodes, set_triplet_classifier, contrastive_ebp), ' + where, 'ms_median': ts[len(ts) // 2], 'ms_min': ts[0], 'type': str(type(s))}))
