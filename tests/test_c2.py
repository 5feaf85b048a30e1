"""the reference's inpainting-game callers (generate_whitebox_saliency.py:79-214) through
xfr_amd.inpainting_game, against golden vectors produced by the reference's OWN caller module
(tests/golden/make_golden_c2.py -> golden_c2.npz): meanEBP over the 65359-way hooked classifier (P = ones), contrastive /
truncated triplet EBP from averaged unit-norm encodings, weighted-subtree EBP with do_mated_similarity_gating=False and
with the ebp_version 8 / 9 / 10 parameterisations (uint8 saliency path).

CPU part: the oracle behind the same caller functions.  GPU part: the HIP engine behind them."""
import os
import types

import numpy as np
import pytest
import torch

import golden_cases as GC
from parity_utils import MAP_RTOL_CONTRAST, assert_map_close, assert_map_close_robust, assert_trace_close, make_backbone, map_metrics
from xfr_amd import inpainting_game as IG
from xfr_amd import synth
from xfr_amd.models import whitebox as WB
from xfr_amd.models.resnet import convert_resnet101v4_image

CPU = torch.device('cpu')


class OracleCaller(object):
    """The slice of the Whitebox surface the generator's callers touch, served by the CPU oracle."""

    def __init__(self, arch, sd, mode, ebp_version=6):
        from oracle import ebp_oracle as O
        self.ow = O.OracleWhitebox(arch, sd, ('hooked', None), mode, ebp_version=ebp_version)
        self.net = types.SimpleNamespace(set_triplet_classifier=self.ow.set_triplet_classifier, num_classes=self.ow.num_classes,
                                         preprocess=lambda im: convert_resnet101v4_image(im.resize((224, 224))).unsqueeze(0))
        self.encode, self.ebp = self.ow.encode, self.ow.ebp

    def convert_from_numpy(self, img):
        return WB.Whitebox.convert_from_numpy(self, img)

    def contrastive_ebp(self, x, k_poschannel, k_negchannel):
        return self.ow.contrastive_ebp(x, k_poschannel, k_negchannel)

    def truncated_contrastive_ebp(self, x, k_poschannel, k_negchannel, percentile=20):
        return self.ow.truncated_contrastive_ebp(x, k_poschannel, k_negchannel, percentile)

    def weighted_subtree_ebp(self, x, k_poschannel, k_negchannel, topk=1, verbose=True, **kw):
        return self.ow.weighted_subtree_ebp(x, k_poschannel, k_negchannel, topk=topk, **kw)


def test_convert_from_numpy_matches_reference_preprocessing():
    """whitebox.py:787-806 on 224x224 inputs (where skimage's resize is the identity): uint8, [0,1] float and [0,255] float
    inputs all land on the reference's tensor = preprocess(PIL(uint8))."""
    im_mates, _, probe = GC.c2_images()
    wb = OracleCaller('stresnet_mini', make_backbone('stresnet_mini', seed=3, num_classes=5)[1], 'norelu')
    want = convert_resnet101v4_image(probe).unsqueeze(0)
    assert torch.equal(wb.convert_from_numpy(probe), want)
    # the float paths go through (x / 255 * 255).astype(uint8), which truncates: reproduce the numpy expression itself
    f = probe.astype(float) / 255
    assert torch.equal(wb.convert_from_numpy(f), convert_resnet101v4_image((f * 255).astype(np.uint8)).unsqueeze(0))
    f255 = probe.astype(np.float32) * 0.999 + 0.1
    assert torch.equal(wb.convert_from_numpy(f255), convert_resnet101v4_image(((f255 / 255) * 255).astype(np.uint8)).unsqueeze(0))
    with pytest.raises(ValueError):
        wb.convert_from_numpy(probe.astype(np.float32) - 300.0)
    out = wb.convert_from_numpy(np.random.RandomState(0).rand(100, 80, 3))      # other sizes: shape only (parity unpinned)
    assert tuple(out.shape) == (1, 3, 224, 224)
    t = torch.zeros((3, 224, 224))
    assert wb.convert_from_numpy(t).shape == (1, 3, 224, 224)                  # additive: tensors pass through
    # preprocess_loader (whitebox.py:808-825), in-memory branch of image_loader: (image, tensor[3,H,W], fn=None)
    items = list(WB.Whitebox.preprocess_loader(wb, [probe, f]))
    assert len(items) == 2 and items[0][2] is None and torch.equal(items[0][1], want[0]) and items[1][0] is f
    # file-name branch (utils.py:86-90): decode, / 255, centre crop (the identity on a 224 x 224 file), then the float path above
    import tempfile
    import PIL.Image
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, 'probe.png')
        PIL.Image.fromarray(probe).save(fn)
        items = list(WB.Whitebox.preprocess_loader(wb, [fn]))
    assert len(items) == 1 and items[0][2] == fn and np.array_equal(items[0][0], f)
    assert torch.equal(items[0][1], convert_resnet101v4_image((f * 255).astype(np.uint8)))
    with pytest.raises(NotImplementedError):
        list(WB.Whitebox.preprocess_loader(wb, [42]))


def _check_r101(wb, gold, mode, key_check):
    im_mates, im_nonmates, probe = GC.c2_images()
    pre = 'c2/r101/%s/' % mode
    key_check(pre + 'mean_ebp', IG.mean_ebp(wb, probe, 'resnetv4_pytorch', 6, wb_device(wb)))
    key_check(pre + 'contrastive', IG.run_contrastive_triplet_ebp(wb, im_mates, im_nonmates, probe, 'resnetv4_pytorch', 6, None, wb_device(wb)))
    key_check(pre + 'truncated20', IG.run_contrastive_triplet_ebp(wb, im_mates, im_nonmates, probe, 'resnetv4_pytorch', 6, 20, wb_device(wb)))


def wb_device(wb):
    return getattr(wb, '_test_device', CPU)


def test_oracle_c2_resnet101_norelu(monkeypatch):
    """The oracle behind the caller functions reproduces the reference's caller outputs: to 1e-5 when the k images are encoded
    one forward at a time like the reference does.  Encoded as one batch the classifier rows move by ~2e-8 and the contrastive
    map by ~1e-3: the averaged mate / non-mate directions of this fixture have cosine 0.9998 (random weights), which is the
    amplification the contrastive tolerance (parity_utils.MAP_RTOL_CONTRAST) is stated for."""
    torch.set_num_threads(8)
    gold = GC.golden('golden_c2')
    bb, sd = make_backbone('stresnet101', seed=0, num_classes=65359)
    assert synth.state_checksum(sd) == str(gold['r101/wsum'])
    wb = OracleCaller('stresnet101', sd, 'norelu')
    monkeypatch.setattr(IG, 'ENCODE_ONE_BY_ONE', True)

    def chk(key, res):
        rel, _ = map_metrics(res, gold[key + '/map'])
        assert rel <= (1e-3 if key.endswith('truncated20') else 1e-5), '%s: %.3e' % (key, rel)
    _check_r101(wb, gold, 'norelu', chk)
    monkeypatch.setattr(IG, 'ENCODE_ONE_BY_ONE', False)
    im_mates, im_nonmates, probe = GC.c2_images()
    res = IG.run_contrastive_triplet_ebp(wb, im_mates, im_nonmates, probe, 'resnetv4_pytorch', 6, None, CPU)
    assert_map_close(res, gold['c2/r101/norelu/contrastive/map'], 'batched encodes', rtol=MAP_RTOL_CONTRAST)
    cm, cn = gold['c2/r101/norelu/cls_mate'][0], gold['c2/r101/norelu/cls_nonmate'][0]
    assert float(cm @ cn / np.linalg.norm(cm) / np.linalg.norm(cn)) > 0.999      # the conditioning the docstring talks about


@pytest.mark.parametrize('key,ver,mode_w', [('c2/mini/v08_all_top8', 8, 'all'), ('c2/mini/v11_all_top8', 11, 'all'),
                                            ('c2/mini/all/nogate_top8', 6, 'all')])
def test_oracle_c2_subtree_variants_mini(key, ver, mode_w):
    torch.set_num_threads(8)
    gold = GC.golden('golden_c2')
    bb, sd = make_backbone('stresnet_mini', seed=3, recipe='mild', num_classes=5)
    im_mates, im_nonmates, probe = GC.c2_images()
    wb = OracleCaller('stresnet_mini', sd, 'norelu', ebp_version=ver)
    wb.ow.with_bias = (ver == 11)                      # whitebox.py:285-289
    if ver == 6:
        x = wb.convert_from_numpy(probe)
        wb.net.set_triplet_classifier(synth.unit_rows(1, 512, seed=1), synth.unit_rows(1, 512, seed=2))
        smap, _, w_valid, k_valid = wb.weighted_subtree_ebp(x, 0, 1, topk=8, do_mated_similarity_gating=False, subtree_mode=mode_w)
        assert sorted(int(k) for k in k_valid) == sorted(int(k) for k in gold[key + '/k_valid'])
        assert np.allclose(sorted(w_valid), sorted(gold[key + '/w_valid']), rtol=1e-4)
        assert map_metrics(smap, gold[key + '/map'])[0] <= 1e-4
    else:
        smap = IG.run_weighted_subtree_triplet_ebp(wb, im_mates, im_nonmates, probe, 'resnetv4_pytorch', mode_w, ver, CPU, topk=8)
        want = gold[key + '/map']
        assert smap.dtype == np.uint8 and want.dtype == np.uint8
        assert np.abs(smap.astype(int) - want.astype(int)).max() <= 1      # uint8 quantisation: at most one level


# ---- the HIP engine -----------------------------------------------------------------------------------------------------
def _engine_wb(arch, bb, mode, dev, ebp_version=None):
    bb.to(dev)
    wb = WB.Whitebox(WB.WhiteboxSTResnet(bb), ebp_version=ebp_version, ebp_subtree_mode=mode)
    wb._test_device = dev
    return wb


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['norelu', 'affineonly_with_prior'])
def test_engine_c2_resnet101(gpu_device, mode):
    """One generator job's three cheap methods on ResNet-101 (65359-way hooked classifier for meanEBP) vs the reference."""
    gold = GC.golden('golden_c2')
    bb, sd = make_backbone('stresnet101', seed=0, num_classes=65359)
    wb = _engine_wb('stresnet101', bb, mode, gpu_device)
    wb.debug_trace = True

    def chk(key, res):
        want = gold[key + '/map']
        if key.endswith('mean_ebp'):
            assert_map_close_robust(res, want, key)
            assert_trace_close(np.asarray(wb.P_trace)[:, 0], wb.P_layername, gold[key + '/trace'], gold[key + '/names'], key)
        elif key.endswith('truncated20'):
            assert_map_close_robust(res, want, key, rtol=MAP_RTOL_CONTRAST)
        else:
            assert_map_close(res, want, key, rtol=MAP_RTOL_CONTRAST)
    im_mates, im_nonmates, probe = GC.c2_images()
    pre = 'c2/r101/%s/' % mode
    chk(pre + 'mean_ebp', IG.mean_ebp(wb, probe, 'resnetv4_pytorch', 6, gpu_device))
    # (1) the encode half of the callers: averaged unit-norm encodings against the reference's
    gm, gn = gold[pre + 'cls_mate'] * 2500.0, gold[pre + 'cls_nonmate'] * 2500.0
    em = IG.mean_encoding(wb, im_mates, gpu_device).cpu().numpy()
    en = IG.mean_encoding(wb, im_nonmates, gpu_device).cpu().numpy()
    assert np.abs(em - gm).max() <= 2e-5 and np.abs(en - gn).max() <= 2e-5
    # (2) the EBP half under the reference's classifier rows (cosine(mate, non-mate) = 0.9998 here: a 2e-8 change of a row
    #     moves the reference's own map by 1e-3, so the maps are compared under the same rows)
    x = wb.convert_from_numpy(probe).to(gpu_device)
    wb.net.set_triplet_classifier(torch.from_numpy(gold[pre + 'cls_mate']), torch.from_numpy(gold[pre + 'cls_nonmate']))
    chk(pre + 'contrastive', wb.contrastive_ebp(x, 0, 1))
    chk(pre + 'truncated20', wb.truncated_contrastive_ebp(x, 0, 1, percentile=20))
    # (3) end to end through the caller, classifier from the engine's own encodings: same map up to that amplification
    res = IG.run_contrastive_triplet_ebp(wb, im_mates, im_nonmates, probe, 'resnetv4_pytorch', 6, None, gpu_device)
    rel, cos = map_metrics(res, gold[pre + 'contrastive/map'])
    assert np.isfinite(res).all() and abs(float(res.sum()) - 1.0) < 1e-4 and cos >= 0.999, (rel, cos)
    got = wb.net._classifier.weight.cpu().numpy() * 2500.0
    assert np.abs(got - np.concatenate((gm, gn))).max() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize('key,ver,mode_w', [('c2/mini/v07_all_top8', 7, 'all'), ('c2/mini/v08_all_top8', 8, 'all'),
                                            ('c2/mini/v09_norelu_top8', 9, 'norelu'), ('c2/mini/v10_norelu_top8', 10, 'norelu'),
                                            ('c2/mini/v11_all_top8', 11, 'all'),
                                            ('c2/mini/v12_affineonly_with_prior_top8', 12, 'affineonly_with_prior')])
def test_engine_c2_subtree_versions(gpu_device, key, ver, mode_w):
    """generate_whitebox_saliency.py:171-194: every ebp_version the generator knows (7 .. 12; 11 also switches with_bias on)
    through run_weighted_subtree_triplet_ebp (uint8 maps)."""
    gold = GC.golden('golden_c2')
    bb, sd = make_backbone('stresnet_mini', seed=3, recipe='mild', num_classes=5)
    wb = _engine_wb('stresnet_mini', bb, 'norelu', gpu_device, ebp_version=ver)
    im_mates, im_nonmates, probe = GC.c2_images()
    smap = IG.run_weighted_subtree_triplet_ebp(wb, im_mates, im_nonmates, probe, 'resnetv4_pytorch', mode_w, ver, gpu_device, topk=8)
    want = gold[key + '/map']
    assert smap.dtype == np.uint8 and smap.shape == want.shape
    d = np.abs(smap.astype(int) - want.astype(int))
    assert d.max() <= 2 and (d > 0).mean() <= 0.02, '%s: max level diff %d, %.2f %% pixels differ' % (key, d.max(), 100 * (d > 0).mean())


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['norelu', 'all'])
def test_engine_c2_subtree_without_similarity_gating(gpu_device, mode):
    """whitebox.py:657-664,692-694: the cross-entropy-gradient gate (do_mated_similarity_gating=False)."""
    gold = GC.golden('golden_c2')
    key = 'c2/mini/%s/nogate_top8' % mode
    bb, sd = make_backbone('stresnet_mini', seed=3, recipe='mild', num_classes=5)
    wb = _engine_wb('stresnet_mini', bb, mode, gpu_device)
    _, _, probe = GC.c2_images()
    x = wb.convert_from_numpy(probe).to(gpu_device)
    wb.net.set_triplet_classifier(synth.unit_rows(1, 512, seed=1), synth.unit_rows(1, 512, seed=2))
    smap, P_valid, w_valid, k_valid = wb.weighted_subtree_ebp(x, 0, 1, topk=8, verbose=False, do_mated_similarity_gating=False,
                                                              subtree_mode=mode)
    assert sorted(int(k) for k in k_valid) == sorted(int(k) for k in gold[key + '/k_valid'])
    assert np.allclose(sorted(w_valid), sorted(gold[key + '/w_valid']), rtol=1e-4)
    assert_map_close(smap, gold[key + '/map'], key)
