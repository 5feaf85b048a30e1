#!/usr/bin/env python
"""Golden vectors for SURVEY.md 8(a) row C2 and for the weighted-subtree parameterisations the generator runs, produced by
the REAL reference's own caller code (python/xfr/inpainting_game/generate_whitebox_saliency.py, imported from
/root/reference):

    mean_ebp(wb, probe_im, ...)                         :207-214   P = ones(1, 65359) through the hooked classifier
    run_contrastive_triplet_ebp(wb, mates, nonmates, probe, ..., truncate_percent)   :79-115
    run_weighted_subtree_triplet_ebp(wb, mates, nonmates, probe, ..., ebp_version)   :119-205  (versions 8, 9, 10)
    Whitebox.weighted_subtree_ebp(..., do_mated_similarity_gating=False)             whitebox.py:657-664,692-694

Usage (build container only):  python tests/golden/make_golden_c2.py   ->  tests/golden/golden_c2.npz
Inputs are the four bundled JPEGs (tests/golden/inputs_jpeg.npz, already 224x224) and mirrored copies of them, handed to
the reference as image_loader would (float64 RGB in [0, 1], xfr/utils.py:88-90) or as uint8; `Whitebox.convert_from_numpy`
(whitebox.py:787-806) then only meets 224x224 inputs, for which skimage.transform.resize is the identity (ref_import.py).
"""
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
from parity_utils import make_backbone  # noqa: E402
from xfr_amd import synth  # noqa: E402
from make_golden import record, ref_net  # noqa: E402

ns = ref_import.load()
import xfr.inpainting_game.generate_whitebox_saliency as G  # noqa: E402  (the reference's caller module)

torch.set_num_threads(int(os.environ.get('XFR_THREADS', '8')))
CPU = torch.device('cpu')


c2_images = GC.c2_images


def spy_encodings(wb, store, key):
    """Record what set_triplet_classifier receives (the averaged, unit-normalised encodings)."""
    orig = wb.net.set_triplet_classifier

    def spy(x_mate, x_nonmate):
        store[key + '/cls_mate'] = x_mate.detach().cpu().numpy().astype(np.float32)
        store[key + '/cls_nonmate'] = x_nonmate.detach().cpu().numpy().astype(np.float32)
        return orig(x_mate, x_nonmate)
    wb.net.set_triplet_classifier = spy
    return orig


def timed(out, key, wb, call):
    t = time.time()
    res = call()
    for k, v in record(wb, res).items():
        out['%s/%s' % (key, k)] = v
    wb._ebp_mode = 'disable'
    print('  %-56s %.1fs  sum=%.6f' % (key, time.time() - t, float(np.sum(res))))


def subtree(out, key, wb, call):
    """Run a weighted-subtree call, recording which layers the reference's layerwise_ebp visited."""
    visited = []
    orig = wb.layerwise_ebp

    def spy(img, k_layer, mode='argmax', k_element=None, k_poschannel=0, mwp=True):
        r = orig(img, k_layer=k_layer, mode=mode, k_element=k_element, k_poschannel=k_poschannel, mwp=mwp)
        visited.append((int(k_layer), int(k_element), float(np.max(r))))
        return r
    wb.layerwise_ebp = spy
    t = time.time()
    res = call()
    wb.layerwise_ebp = orig
    wb._ebp_mode = 'disable'
    if isinstance(res, tuple):
        smap, P_valid, w_valid, k_valid = res
        out[key + '/k_valid'] = np.array([int(k) for k in k_valid])
        out[key + '/w_valid'] = np.array([float(w) for w in w_valid])
        out[key + '/P_valid'] = np.stack([np.asarray(p) for p in P_valid])
    else:
        smap = res
    out[key + '/map'] = np.asarray(smap)
    out[key + '/visit_layer'] = np.array([v[0] for v in visited])
    out[key + '/visit_elem'] = np.array([v[1] for v in visited])
    print('  %-56s %.1fs  dtype %s  sum=%.6f' % (key, time.time() - t, np.asarray(smap).dtype, float(np.sum(smap))))


def main(sections):
    """sections: any of 'r101', 'mini_nogate', 'mini_versions' (default: all).  Cases of sections that are not regenerated are
    kept from the existing golden_c2.npz."""
    path = os.path.join(HERE, 'golden_c2.npz')
    out = dict(np.load(path)) if (sections and os.path.exists(path)) else {}
    sections = sections or ['r101', 'mini_nogate', 'mini_versions']
    im_mates, im_nonmates, probe_im = c2_images()

    # ---- ResNet-101: the three cheap methods of one generator job --------------------------------------------------
    if 'r101' in sections:
        NC = 65359
        bb, sd = make_backbone('stresnet101', seed=0, recipe='mild', num_classes=NC)
        out['r101/wsum'] = np.array(synth.state_checksum(sd))
        for mode in ('norelu', 'affineonly_with_prior'):
            wbn = ref_net('stresnet101', sd, NC)
            wb = ns.whitebox.Whitebox(wbn, ebp_subtree_mode=mode)
            pre = 'c2/r101/%s/' % mode
            timed(out, pre + 'mean_ebp', wb, lambda: G.mean_ebp(wb, probe_im, 'resnetv4_pytorch', 6, CPU))
            spy_encodings(wb, out, pre[:-1])
            timed(out, pre + 'contrastive', wb,
                  lambda: G.run_contrastive_triplet_ebp(wb, im_mates, im_nonmates, probe_im, 'resnetv4_pytorch', 6, None, CPU))
            timed(out, pre + 'truncated20', wb,
                  lambda: G.run_contrastive_triplet_ebp(wb, im_mates, im_nonmates, probe_im, 'resnetv4_pytorch', 6, 20, CPU))

    # ---- mini STR-ResNet: weighted-subtree parameterisations -----------------------------------------------------------
    bb, sd = make_backbone('stresnet_mini', seed=3, recipe='mild', num_classes=5)
    out['mini/wsum'] = np.array(synth.state_checksum(sd))
    # (a) do_mated_similarity_gating=False through the method itself, float32 maps (ebp_version 6)
    if 'mini_nogate' in sections:
        for mode in ('norelu', 'all'):
            wbn = ref_net('stresnet_mini', sd, 5)
            wb = ns.whitebox.Whitebox(wbn, ebp_subtree_mode=mode)
            x = wb.convert_from_numpy(probe_im)
            em = torch.from_numpy(synth.unit_rows(1, 512, seed=1).numpy())
            en = torch.from_numpy(synth.unit_rows(1, 512, seed=2).numpy())
            wbn.set_triplet_classifier(em, en)
            subtree(out, 'c2/mini/%s/nogate_top8' % mode, wb,
                    lambda: wb.weighted_subtree_ebp(x, 0, 1, topk=8, verbose=False, do_mated_similarity_gating=False, subtree_mode=mode))
    # (b) the generator's own caller with every ebp_version it knows, 7 .. 12 (uint8 saliency path: whitebox.py:451-454,726-727;
    #     11 additionally turns with_bias on: whitebox.py:285-289); the weighted mode is the one the generator's docstring pairs
    #     with the version (generate_whitebox_saliency.py:145-169)
    if 'mini_versions' in sections:
        for ver, mode_w in ((7, 'all'), (8, 'all'), (9, 'norelu'), (10, 'norelu'), (11, 'all'), (12, 'affineonly_with_prior')):
            wbn = ref_net('stresnet_mini', sd, 5)
            wb = ns.whitebox.Whitebox(wbn, ebp_version=ver, ebp_subtree_mode='norelu')       # create_wbnet.py:51-66
            key = 'c2/mini/v%02d_%s_top8' % (ver, mode_w)
            if key + '/map' in out:
                continue
            spy_encodings(wb, out, key)
            subtree(out, key, wb,
                    lambda: G.run_weighted_subtree_triplet_ebp(wb, im_mates, im_nonmates, probe_im, 'resnetv4_pytorch', mode_w, ver, CPU, topk=8))
    np.savez_compressed(path, **out)
    print('done')


if __name__ == '__main__':
    main(sys.argv[1:])
