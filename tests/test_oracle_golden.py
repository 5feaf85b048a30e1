"""The CPU oracle against the golden vectors captured from the real reference (tests/golden/make_golden.py).

On the machine that produced the vectors the oracle reproduces them bit-for-bit (it dispatches to the same ATen
kernels); across machines / thread counts MKLDNN may re-associate the convolution sums, so the assertion is
max|d|/max <= 1e-5 on maps and 1e-5 relative on every per-firing P sum."""
import numpy as np
import pytest
import torch

import golden_cases as GC
from parity_utils import make_backbone, map_metrics
from xfr_amd import synth

ORACLE_TOL = 1e-5
# ATen's CPU convolutions are not run-to-run deterministic for strided 1x1 convs (thread partitioning): ResNet-50's maps
# move by ~1e-7, which the contrastive subtraction and the percentile mask of the truncated variant amplify
ORACLE_TOL_CONTRAST = 1e-4
ORACLE_TOL_TRUNCATED = 1e-3


def check(key, res, trace, gold):
    want = gold[key + '/map']
    rel, cos = map_metrics(res, want)
    tol = ORACLE_TOL_TRUNCATED if key.endswith('truncated') else (ORACLE_TOL_CONTRAST if key.endswith('contrastive') else ORACLE_TOL)
    assert rel <= tol, '%s: map max|d|/max = %.3e' % (key, rel)
    sums, names = trace
    gsum, gnames = gold[key + '/trace'], [str(n) for n in gold[key + '/names']]
    assert names == gnames, '%s: firing order differs' % key
    err = np.abs(sums - gsum) / np.maximum(np.abs(gsum), 1e-300)
    assert err.max() <= ORACLE_TOL, '%s: trace rel err %.3e at firing %d (%s)' % (key, err.max(), int(err.argmax()), names[int(err.argmax())])


@pytest.mark.parametrize('recipe', ['mild', 'harsh'])
@pytest.mark.parametrize('mode', ['affineonly_with_prior', 'norelu', 'all', 'affineonly'])
def test_mini_resnet_all_modes(recipe, mode):
    torch.set_num_threads(8)
    gold = GC.golden('golden_mini')
    bb, sd = make_backbone('stresnet_mini', seed=3, recipe=recipe, num_classes=5)
    assert synth.state_checksum(sd) == str(gold['mini/%s/wsum' % recipe]), 'synthetic weight generator drifted'
    GC.replay(GC.oracle_subject('stresnet_mini', sd, mode), GC.mini_cases(recipe, mode), gold, check)


def test_resnet101_triplet_contrastive_demo():
    """demo/test_whitebox.py:124-133 on the bundled JPEGs + the bench-style synthetic triplet (default mode)."""
    torch.set_num_threads(8)
    gold = GC.golden('golden_r101')
    bb, sd = make_backbone('stresnet101', seed=0, num_classes=65359)
    assert synth.state_checksum(sd) == str(gold['r101/wsum'])
    cases = GC.r101_cases('affineonly_with_prior', which=['triplet/contrastive', 'synthetic/contrastive', 'hooked/ebp'])
    GC.replay(GC.oracle_subject('stresnet101', sd, 'affineonly_with_prior'), cases, gold, check)


def test_resnet50_128_truncated_norelu():
    torch.set_num_threads(8)
    gold = GC.golden('golden_r50')
    bb, sd = make_backbone('resnet50_128', seed=0)
    assert synth.state_checksum(sd) == str(gold['r50/wsum'])
    cases = GC.r50_cases('norelu', which=['triplet/truncated', 'triplet/ebp'])
    GC.replay(GC.oracle_subject('resnet50_128', sd, 'norelu'), cases, gold, check)


def test_lightcnn_ebp_affineonly():
    torch.set_num_threads(8)
    gold = GC.golden('golden_lcnn')
    bb, sd = make_backbone('lightcnn29v2', seed=0, num_classes=80013)
    assert synth.state_checksum(sd) == str(gold['lcnn/wsum'])
    GC.replay(GC.oracle_subject('lightcnn29v2', sd, 'affineonly'), GC.lcnn_cases('affineonly'), gold, check)
    cases = GC.lcnn_cases('affineonly_with_prior', which=['triplet/contrastive'])
    GC.replay(GC.oracle_subject('lightcnn29v2', sd, 'affineonly_with_prior'), cases, gold, check)


def test_weighted_subtree_and_layerwise_mini():
    """"next" row (whitebox.py:561-581, 647-737): the oracle's restatement against the reference's own outputs."""
    torch.set_num_threads(8)
    from oracle import ebp_oracle as O
    from parity_utils import make_images
    g = GC.golden('golden_subtree_mini')
    bb, sd = make_backbone('stresnet_mini', seed=3, recipe='mild', num_classes=5)
    x = make_images('stresnet_mini', 1, seed=5)
    ow = O.OracleWhitebox('stresnet_mini', sd, ('hooked', None), 'norelu')
    ow.set_triplet_classifier(synth.unit_rows(1, 512, seed=1) / 2500, synth.unit_rows(1, 512, seed=2) / 2500)
    smap, P_valid, w_valid, k_valid = ow.weighted_subtree_ebp(x, 0, 1, topk=8, subtree_mode='norelu')
    key = 'mini/norelu/top8'
    assert [int(k) for k in k_valid] == [int(k) for k in g[key + '/k_valid']]
    assert np.allclose(np.array(w_valid), g[key + '/w_valid'], rtol=ORACLE_TOL)
    assert map_metrics(smap, g[key + '/map'])[0] <= ORACLE_TOL
    for k in (5, 20, 40):
        got = ow.layerwise_ebp(x, k_layer=k, mode='argmax', k_poschannel=0, mwp=True)
        want = g['mini/norelu/layerwise_argmax_%d' % k]
        assert np.abs(got - want).max() <= ORACLE_TOL * max(np.abs(want).max(), 1e-30)



@pytest.mark.parametrize('arch,tag,mode,nc', [('resnet50_128', 'r50', 'norelu', None), ('lightcnn29v2', 'lcnn', 'all', 7)])
def test_well_conditioned_contrastive(arch, tag, mode, nc):
    """tests/golden/make_golden_synth.py: independent random classifier rows (the contrast does not cancel)."""
    torch.set_num_threads(8)
    gold = GC.golden('golden_synth')
    bb, sd = make_backbone(arch, seed=0, num_classes=nc)
    assert synth.state_checksum(sd) == str(gold[tag + '/wsum'])
    GC.replay(GC.oracle_subject(arch, sd, mode), GC.synth_cases(arch, tag, mode), gold, check)


def test_layerwise_contrastive_ebp_mini():
    """whitebox.py:584-645 (deprecated by the reference; restated for completeness): every mode on three layers, against the reference's maps."""
    import warnings
    torch.set_num_threads(8)
    from oracle import ebp_oracle as O
    from parity_utils import make_images
    g = GC.golden('golden_lwc_mini')
    bb, sd = make_backbone('stresnet_mini', seed=3, recipe='mild', num_classes=5)
    x = make_images('stresnet_mini', 1, seed=5)
    ow = O.OracleWhitebox('stresnet_mini', sd, ('hooked', None), 'norelu')
    ow.set_triplet_classifier(synth.unit_rows(1, 512, seed=1) / 2500, synth.unit_rows(1, 512, seed=2) / 2500)
    ow.ebp(x, ow._onehot(x, 0))
    shapes = [p for p in ow.P]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for k in [int(v) for v in g['mini/norelu/layers']]:
            kel = int(g['mini/norelu/k_element_%d' % k])
            for mode in ('copy', 'mean', 'product', 'argmax', 'argmax_product', 'percentile', 'percentile_argmax', 'elementwise'):
                got = ow.layerwise_contrastive_ebp(x, 0, 1, k_layer=k, mode=mode, percentile=80, k_element=kel, gradlayer=shapes, mwp=True)
                want = g['mini/norelu/%s_%d' % (mode, k)]
                assert np.abs(np.asarray(got) - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-30), (mode, k)

