#!/usr/bin/env python
"""Golden vectors for the "next" row layerwise_ebp / weighted_subtree_ebp (whitebox.py:561-581, 647-737), produced by the
REAL reference.  Usage: python tests/golden/make_golden_subtree.py [mini|r101]   ->  tests/golden/golden_subtree_<x>.npz
Stored per case: final map, the selected subtree layer indices and weights, the per-subtree maps, and -- captured by
wrapping the reference's own layerwise_ebp -- the order in which layers were visited and the element chosen in each."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
import golden_cases as GC  # noqa: E402
from parity_utils import make_backbone, make_images  # noqa: E402
from xfr_amd import synth  # noqa: E402
from make_golden import ref_net  # noqa: E402

ns = ref_import.load()
torch.set_num_threads(int(os.environ.get("XFR_THREADS", "8")))


def run(out, key, wb, x, topk, mode, **kw):
    visited = []
    orig = wb.layerwise_ebp

    def spy(img, k_layer, mode='argmax', k_element=None, k_poschannel=0, mwp=True):
        r = orig(img, k_layer=k_layer, mode=mode, k_element=k_element, k_poschannel=k_poschannel, mwp=mwp)
        visited.append((int(k_layer), int(k_element), float(np.max(r))))
        return r
    wb.layerwise_ebp = spy
    t = time.time()
    smap, P_valid, w_valid, k_valid = wb.weighted_subtree_ebp(x, 0, 1, topk=topk, verbose=False, subtree_mode=mode, **kw)
    wb.layerwise_ebp = orig
    wb._ebp_mode = 'disable'
    out[key + '/map'] = np.asarray(smap, dtype=np.float32)
    out[key + '/k_valid'] = np.array([int(k) for k in k_valid])
    out[key + '/w_valid'] = np.array([float(w) for w in w_valid])
    out[key + '/P_valid'] = np.stack([np.asarray(p, dtype=np.float32) for p in P_valid])
    out[key + '/visit_layer'] = np.array([v[0] for v in visited])
    out[key + '/visit_elem'] = np.array([v[1] for v in visited])
    out[key + '/visit_max'] = np.array([v[2] for v in visited])
    print('  %-50s %.1fs  valid %s' % (key, time.time() - t, list(out[key + '/k_valid'])))


def main(which):
    out = {}
    if which == 'mini':
        bb, sd = make_backbone('stresnet_mini', seed=3, recipe='mild', num_classes=5)
        x = make_images('stresnet_mini', 1, seed=5)
        for mode in ('norelu', 'affineonly_with_prior', 'all'):
            wbn = ref_net('stresnet_mini', sd, 5)
            wb = ns.whitebox.Whitebox(wbn, ebp_subtree_mode=mode)
            wbn.set_triplet_classifier(synth.unit_rows(1, 512, seed=1) / 2500, synth.unit_rows(1, 512, seed=2) / 2500)
            run(out, 'mini/%s/top8' % mode, wb, x, 8, mode)
            if mode == 'norelu':
                run(out, 'mini/%s/top3max' % mode, wb, x, 3, mode, do_max_subtree=True)
                for k in (5, 20, 40):
                    r = wb.layerwise_ebp(x, k_layer=k, mode='argmax', k_poschannel=0, mwp=True)
                    wb._ebp_mode = 'disable'
                    out['mini/%s/layerwise_argmax_%d' % (mode, k)] = np.asarray(r, dtype=np.float32)
    else:
        bb, sd = make_backbone('stresnet101', seed=0, recipe='mild', num_classes=65359)
        gold = GC.golden('golden_r101')
        x_demo, x_probe, x_non, x_mate = GC.net_inputs('stresnet101')
        mode = 'norelu'
        wbn = ref_net('stresnet101', sd, 65359)
        wb = ns.whitebox.Whitebox(wbn, ebp_subtree_mode=mode)
        em = torch.from_numpy(gold['r101/norelu/enc_mate'])
        en = torch.from_numpy(gold['r101/norelu/enc_nonmate'])
        wbn.set_triplet_classifier((1.0 / 2500.0) * em, (1.0 / 2500.0) * en)
        run(out, 'r101/%s/top32' % mode, wb, x_probe, 32, mode)      # test_whitebox.py:173-199 shape, topk 32 as in the eval
    np.savez_compressed(os.path.join(HERE, 'golden_subtree_%s.npz' % which), **out)
    print('done')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'mini')
