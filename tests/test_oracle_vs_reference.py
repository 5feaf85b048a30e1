"""Live cross-check of the oracle against the real reference, P tensor by P tensor.  Only runs where /root/reference
exists (the build container); on the GPU box the committed golden vectors stand in for it."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import ref_import  # noqa: E402
from parity_utils import make_backbone, make_images  # noqa: E402
from xfr_amd import synth  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_import.available(), reason='reference checkout not present')


@pytest.mark.parametrize('mode', ['affineonly_with_prior', 'norelu', 'all', 'affineonly'])
def test_every_P_tensor_matches_reference(mode):
    from oracle import ebp_oracle as O
    ns = ref_import.load()
    torch.set_num_threads(8)
    bb, sd = make_backbone('stresnet_mini', seed=9, recipe='harsh', num_classes=5)
    net = ns.resnet.ResNet(ns.resnet.Bottleneck, [1, 1, 1, 1], mode='encode', num_classes=5)
    net.load_state_dict(sd, strict=True)
    net.eval()
    wbn = ns.whitebox.WhiteboxSTResnet(net)
    wb = ns.whitebox.Whitebox(wbn, ebp_subtree_mode=mode)
    ow = O.OracleWhitebox('stresnet_mini', sd, ('hooked', None), mode)
    x = make_images('stresnet_mini', 1, seed=11)
    for triplet in (False, True):
        if triplet:
            xm, xn = synth.unit_rows(1, 512, seed=1) / 2500, synth.unit_rows(1, 512, seed=2) / 2500
            wbn.set_triplet_classifier(xm, xn)
            ow.set_triplet_classifier(xm, xn)
        C = 2 if triplet else 5
        Pn = torch.zeros(1, C)
        Pn[0, 1] = 1
        wb.ebp(x, Pn, mwp=True)
        Pref = [p.detach().clone() for p in wb.P]
        names = [n.split('(')[0] for n in wb.P_layername]
        wb._ebp_mode = 'disable'
        ow.ebp(x, Pn, mwp=True)
        assert names == ow.P_layername
        for i, (a, b) in enumerate(zip(Pref, ow.P)):
            assert torch.equal(a, b) or float((a - b).abs().max()) <= 1e-6 * float(a.abs().max()), (i, names[i])


def test_with_bias_matches_reference():
    """ebp_version 11 / with_bias=True: biases (and BatchNorm beta) are rectified in the positive pass (whitebox.py:321-324)."""
    from oracle import ebp_oracle as O
    ns = ref_import.load()
    torch.set_num_threads(8)
    bb, sd = make_backbone('stresnet_mini', seed=9, recipe='mild', num_classes=5)
    net = ns.resnet.ResNet(ns.resnet.Bottleneck, [1, 1, 1, 1], mode='encode', num_classes=5)
    net.load_state_dict(sd, strict=True)
    net.eval()
    wb = ns.whitebox.Whitebox(ns.whitebox.WhiteboxSTResnet(net), ebp_subtree_mode='all', with_bias=True)
    ow = O.OracleWhitebox('stresnet_mini', sd, ('hooked', None), 'all', with_bias=True)
    x = make_images('stresnet_mini', 1, seed=11)
    Pn = torch.zeros(1, 5)
    Pn[0, 3] = 1
    wb.ebp(x, Pn, mwp=True)
    Pref = [p.detach().clone() for p in wb.P]
    wb._ebp_mode = 'disable'
    ow.ebp(x, Pn, mwp=True)
    for i, (a, b) in enumerate(zip(Pref, ow.P)):
        assert torch.equal(a, b) or float((a - b).abs().max()) <= 1e-6 * float(a.abs().max()), i


def test_uint8_saliency_path_matches_reference():
    """ebp_version != 6: _mwp_to_saliency goes through uint8 + PIL GaussianBlur (whitebox.py:451-454)."""
    from oracle import ebp_oracle as O
    ns = ref_import.load()
    torch.set_num_threads(8)
    bb, sd = make_backbone('stresnet_mini', seed=9, recipe='mild', num_classes=5)
    net = ns.resnet.ResNet(ns.resnet.Bottleneck, [1, 1, 1, 1], mode='encode', num_classes=5)
    net.load_state_dict(sd, strict=True)
    net.eval()
    wbn = ns.whitebox.WhiteboxSTResnet(net)
    wb = ns.whitebox.Whitebox(wbn, ebp_version=5)
    ow = O.OracleWhitebox('stresnet_mini', sd, ('hooked', None), 'affineonly_with_prior', ebp_version=5)
    xm, xn = synth.unit_rows(1, 512, seed=1) / 2500, synth.unit_rows(1, 512, seed=2) / 2500
    wbn.set_triplet_classifier(xm, xn)
    ow.set_triplet_classifier(xm, xn)
    x = make_images('stresnet_mini', 1, seed=11)
    a = wb.contrastive_ebp(x, 0, 1)
    wb._ebp_mode = 'disable'
    b = ow.contrastive_ebp(x, 0, 1)
    assert a.dtype == np.uint8 and b.dtype == np.uint8 and np.array_equal(a, b)
