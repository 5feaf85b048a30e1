#!/usr/bin/env python
"""Whitebox.P_layername of the REAL reference, in full (whitebox.py:393: str(module) of the hooked module of every firing), for the three
backbones, hooked and triplet classifier.  The strings depend on the layer program only (not on weights or inputs), so one plain ebp() per case
is enough.  Run in the build container only:   python tests/golden/make_golden_names.py   ->   tests/golden/golden_layernames.npz

Note: the strings are torch's reprs of THIS image's torch (2.10); the reference pinned torch 1.3, whose reprs of the same modules differ in
places (e.g. track_running_stats) -- a drop-in user sees what their own torch prints, and so do we: xfr_amd builds the strings by instantiating
the same torch modules (xfr_amd/models/_backbone.py layer_reprs).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, HERE)

import ref_import  # noqa: E402
from make_golden import ref_net  # noqa: E402
from parity_utils import make_backbone, make_images  # noqa: E402
from xfr_amd import synth  # noqa: E402

ns = ref_import.ns if hasattr(ref_import, 'ns') else ref_import.load()
torch.set_num_threads(8)


def main():
    out = {}
    for arch, ncls in (('stresnet_mini', 5), ('stresnet101', 7), ('resnet50_128', None), ('lightcnn29v2', 7)):
        bb, sd = make_backbone(arch, seed=1, num_classes=ncls)
        net = ref_net(arch, sd, ncls)
        wb = ns.whitebox.Whitebox(net, ebp_subtree_mode='affineonly_with_prior')
        x = make_images(arch, 1, seed=3)
        if arch != 'resnet50_128':
            P = torch.zeros((1, ncls))
            P[0, 0] = 1.0
            wb.ebp(x, P)
            out[arch + '/hooked'] = np.array(list(wb.P_layername))
        D = {'stresnet_mini': 512, 'stresnet101': 512, 'resnet50_128': 128, 'lightcnn29v2': 256}[arch]
        net.set_triplet_classifier(synth.unit_rows(1, D, seed=1) / 2500, synth.unit_rows(1, D, seed=2) / 2500)
        wb.contrastive_ebp(x, 0, 1)
        out[arch + '/triplet'] = np.array(list(wb.P_layername))
        print(arch, {k: len(v) for k, v in out.items() if k.startswith(arch)}, flush=True)
    out['torch_version'] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(HERE, 'golden_layernames.npz'), **out)


if __name__ == '__main__':
    main()
