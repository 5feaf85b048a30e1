#!/usr/bin/env python
"""Golden vectors for layerwise_contrastive_ebp (whitebox.py:584-645; deprecated by the reference, kept for drop-in completeness), produced by the
REAL reference on the [1,1,1,1] STR-ResNet: every mode x a few layers, subtree modes 'norelu' and 'affineonly_with_prior'.
Usage: python tests/golden/make_golden_lwc.py  ->  tests/golden/golden_lwc_mini.npz"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from parity_utils import make_backbone, make_images  # noqa: E402
from xfr_amd import synth  # noqa: E402
from make_golden import ref_net  # noqa: E402

ns = ref_import.load()
torch.set_num_threads(8)
MODES = ['copy', 'mean', 'product', 'argmax', 'argmax_product', 'percentile', 'percentile_argmax', 'elementwise']
LAYERS = [6, 23, 41]


def main():
    warnings.simplefilter('ignore')
    out = {}
    bb, sd = make_backbone('stresnet_mini', seed=3, recipe='mild', num_classes=5)
    x = make_images('stresnet_mini', 1, seed=5)
    for sub in ('norelu', 'affineonly_with_prior'):
        wbn = ref_net('stresnet_mini', sd, 5)
        wb = ns.whitebox.Whitebox(wbn, ebp_subtree_mode=sub)
        wbn.set_triplet_classifier(synth.unit_rows(1, 512, seed=1) / 2500, synth.unit_rows(1, 512, seed=2) / 2500)
        P0 = torch.zeros((1, 2)); P0[0][0] = 1.0
        P1 = torch.zeros((1, 2)); P1[0][1] = 1.0
        wb.ebp(x, P0); Pm = [p.detach().clone() for p in wb.P]
        wb.ebp(x, P1); Pn = [p.detach().clone() for p in wb.P]
        wb._ebp_mode = 'disable'
        # layers whose contrast is not identically zero (the two seeds differ only in sign at many hooks of a random network), spread over the depth
        live = [k for k in range(len(Pm) - 1) if float(torch.relu(Pm[k] - Pn[k]).max()) > 0]
        layers = [live[0], live[len(live) // 2], live[-1]] if len(live) >= 3 else LAYERS
        out['mini/%s/layers' % sub] = np.array(layers)
        print(sub, 'layers with a non-zero contrast:', len(live), 'of', len(Pm), '->', layers)
        for k in layers:
            kel = int(torch.argmax(torch.relu(Pm[k] - Pn[k]).flatten()))
            out['mini/%s/k_element_%d' % (sub, k)] = np.array(kel)
            for mode in MODES:
                r = wb.layerwise_contrastive_ebp(x, 0, 1, k_layer=k, mode=mode, percentile=80, k_element=kel, gradlayer=Pm, mwp=True)
                wb._ebp_mode = 'disable'
                r = np.asarray(r, dtype=np.float32)
                assert np.isfinite(r).all()
                out['mini/%s/%s_%d' % (sub, mode, k)] = r
                print('%-22s %-18s k=%2d sum %.4e max %.4e' % (sub, mode, k, float(r.sum()), float(r.max())))
    np.savez_compressed(os.path.join(HERE, 'golden_lwc_mini.npz'), **out)


if __name__ == '__main__':
    main()
