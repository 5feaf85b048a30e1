#!/usr/bin/env python
"""Fixture for bench.py's output check: sample 0 of the synthetic ResNet-101 batch bench.py times (seed 1234, rank 0, batch 32)
through the REAL reference -- encode(mate), encode(nonmate), set_triplet_classifier(./2500), contrastive_ebp(probe, 0, 1)
(demo/test_whitebox.py:124-133), mode affineonly_with_prior -- and the same for ResNet-50-128d truncated (batch 64, norelu).
Usage (build container):  python tests/golden/make_golden_bench.py  ->  tests/golden/golden_bench.npz"""
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
from parity_utils import R50_MEAN, make_backbone  # noqa: E402
from xfr_amd import synth  # noqa: E402
from xfr_amd.models import resnet as xresnet  # noqa: E402
from make_golden import ref_net  # noqa: E402

ns = ref_import.load()
torch.set_num_threads(8)


def case(out, key, arch, mode, B, mean, pct):
    bb, sd = make_backbone(arch, seed=0, recipe='mild', num_classes=2 if arch == 'stresnet101' else None)
    imgs = synth.bench_images(B, (3, 224, 224), seed=1234, mean=mean)          # what bench.py builds for rank 0
    mate, nonmate, probe = imgs[0:1], imgs[B:B + 1], imgs[2 * B:2 * B + 1]
    wbn = ref_net(arch, sd, 2)
    wb = ns.whitebox.Whitebox(wbn, ebp_subtree_mode=mode)
    em, en = wbn.encode(mate).detach(), wbn.encode(nonmate).detach()
    wbn.set_triplet_classifier((1.0 / 2500.0) * em, (1.0 / 2500.0) * en)
    m = wb.contrastive_ebp(probe, 0, 1) if pct is None else wb.truncated_contrastive_ebp(probe, 0, 1, pct)
    assert np.isfinite(m).all()
    out[key + '/map'] = np.asarray(m, dtype=np.float32)
    out[key + '/enc_mate'] = em.numpy()
    out[key + '/enc_nonmate'] = en.numpy()
    out[key + '/wsum'] = np.array(synth.state_checksum(sd))
    c = float(torch.nn.functional.cosine_similarity(em, en).item())
    print('%s: sum %.6f, cosine(mate, nonmate) %.6f' % (key, float(m.sum()), c))


def main():
    out = {}
    case(out, 'bench/r101', 'stresnet101', 'affineonly_with_prior', 32, xresnet.MEAN_RGB, None)
    case(out, 'bench/r50', 'resnet50_128', 'norelu', 64, R50_MEAN, 20)
    np.savez_compressed(os.path.join(HERE, 'golden_bench.npz'), **out)


if __name__ == '__main__':
    main()
