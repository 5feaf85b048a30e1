#!/usr/bin/env python
"""Well-conditioned contrastive fixtures for ResNet-50-128d and Light-CNN-29v2 (round 5), through the REAL reference.

The triplet cases of make_golden.py install classifier rows taken from two face encodings; under seeded random weights those are nearly
parallel (cosine 0.9998), the two sweeps of contrastive_ebp nearly cancel, and the GPU tests hold such maps to 5e-3.  Here the rows are two
independent random unit vectors / 2500 (synth.unit_rows seeds 1 and 2 -- what `r101/.../synthetic/contrastive` of make_golden.py uses for
ResNet-101): the contrast is well conditioned, and the engine is held to SURVEY.md section 8c's 1e-3 on these.  Probes: a synthetic smooth
image per backbone.  Plus the Light-CNN row of bench.py's output check: image 0 of its batch (seed 1234) through the 80013-way hooked
classifier, mode 'affineonly', class 0 (demo/test_whitebox.py:202-254 shape).

Usage (build container):  python tests/golden/make_golden_synth.py  ->  tests/golden/golden_synth.npz"""
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
from parity_utils import emb_dim, make_backbone, make_images  # noqa: E402
from xfr_amd import synth  # noqa: E402
from make_golden import ref_net, run_case  # noqa: E402

ns = ref_import.load()
torch.set_num_threads(8)


def synth_case(out, arch, tag, mode, num_classes):
    bb, sd = make_backbone(arch, seed=0, recipe='mild', num_classes=num_classes)
    out['%s/wsum' % tag] = np.array(synth.state_checksum(sd))
    wbn = ref_net(arch, sd, num_classes)
    wb = ns.whitebox.Whitebox(wbn, ebp_subtree_mode=mode)
    x = make_images(arch, 1, seed=77, smooth=True)
    D = emb_dim(arch)
    wbn.set_triplet_classifier(synth.unit_rows(1, D, seed=1) / 2500, synth.unit_rows(1, D, seed=2) / 2500)
    run_case(out, '%s/%s/synthetic/contrastive' % (tag, mode), wb, lambda w: w.contrastive_ebp(x, 0, 1))
    run_case(out, '%s/%s/synthetic/truncated' % (tag, mode), wb, lambda w: w.truncated_contrastive_ebp(x, 0, 1, 20))


def lcnn_bench_row(out):
    bb, sd = make_backbone('lightcnn29v2', seed=0, recipe='mild', num_classes=80013)
    xs = synth.synth_images(128, (1, 128, 128), seed=1234, scale255=False)          # bench.py, rank 0
    wbn = ref_net('lightcnn29v2', sd, 80013)
    wb = ns.whitebox.Whitebox(wbn, ebp_subtree_mode='affineonly')
    P = torch.zeros((1, 80013))
    P[0, 0] = 1.0
    m = wb.ebp(xs[0:1], P)
    assert np.isfinite(m).all()
    out['bench/lcnn/map'] = np.asarray(m, dtype=np.float32)
    out['bench/lcnn/wsum'] = np.array(synth.state_checksum(sd))
    print('bench/lcnn: sum %.6f' % float(m.sum()))


def main():
    out = {}
    for mode in ('norelu', 'affineonly_with_prior'):
        synth_case(out, 'resnet50_128', 'r50', mode, None)
    for mode in ('affineonly_with_prior', 'all'):
        synth_case(out, 'lightcnn29v2', 'lcnn', mode, 7)
    lcnn_bench_row(out)
    np.savez_compressed(os.path.join(HERE, 'golden_synth.npz'), **out)


if __name__ == '__main__':
    main()
