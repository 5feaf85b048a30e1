"""GPU tests of the "next" row: layerwise_ebp / weighted_subtree_ebp (whitebox.py:561-581, 647-737) against golden vectors
produced by the real reference (tests/golden/make_golden_subtree.py) and against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

import golden_cases as GC
from parity_utils import assert_map_close, assert_map_close_robust, make_backbone, make_images, map_metrics
from xfr_amd import synth

pytestmark = pytest.mark.gpu


def _mini(mode):
    bb, sd = make_backbone('stresnet_mini', seed=3, recipe='mild', num_classes=5)
    subj = GC.engine_subject('stresnet_mini', bb, mode)
    subj.wb.debug_trace = False
    subj.set_cls(synth.unit_rows(1, 512, seed=1) / 2500, synth.unit_rows(1, 512, seed=2) / 2500)
    return subj, sd, make_images('stresnet_mini', 1, seed=5)


@pytest.mark.parametrize('mode', ['norelu', 'affineonly_with_prior', 'all'])
def test_weighted_subtree_mini_golden(gpu_device, mode):
    g = GC.golden('golden_subtree_mini')
    subj, sd, x = _mini(mode)
    key = 'mini/%s/top8' % mode
    smap, P_valid, w_valid, k_valid = subj.wb.weighted_subtree_ebp(x, 0, 1, topk=8, verbose=False, subtree_mode=mode)
    assert [int(k) for k in k_valid] == [int(k) for k in g[key + '/k_valid']]
    assert np.allclose(np.array(w_valid), g[key + '/w_valid'], rtol=1e-4, atol=0)
    for a, b in zip(P_valid, g[key + '/P_valid']):
        assert_map_close_robust(a, b, key + ' subtree map')
    assert_map_close_robust(smap, g[key + '/map'], key)
    assert abs(float(np.sum(smap)) - 1.0) < 1e-4


def test_weighted_subtree_max_variant_and_visit_elements(gpu_device):
    g = GC.golden('golden_subtree_mini')
    subj, sd, x = _mini('norelu')
    key = 'mini/norelu/top3max'
    smap, P_valid, w_valid, k_valid = subj.wb.weighted_subtree_ebp(x, 0, 1, topk=3, verbose=False, subtree_mode='norelu',
                                                                  do_max_subtree=True)
    assert [int(k) for k in k_valid] == [int(k) for k in g[key + '/k_valid']]
    assert_map_close_robust(smap, g[key + '/map'], key)
    # the element the reference picked in every layer (argmax of (dmate >= 0) * (-dnonmate), whitebox.py:690)
    eng = subj.wb._engine(1)
    C = 2
    e0 = torch.zeros((1, C)); e0[0][0] = 1
    e1 = torch.zeros((1, C)); e1[0][1] = 1
    st, s0 = subj.wb.net.seed_for(e0, 1)
    _, s1 = subj.wb.net.seed_for(e1, 1)
    w, idx = eng.subtree_weights(x, st, torch.stack((s0, s1), 0))
    layers, elems = g[key + '/visit_layer'], g[key + '/visit_elem']
    same = sum(int(idx[int(k), 0]) == int(e) for k, e in zip(layers, elems))
    assert same >= 0.95 * len(layers), '%d of %d argmax elements agree' % (same, len(layers))
    # visiting order = ascending layer weight
    order = np.argsort(w[:, 0])
    pos = {int(k): i for i, k in enumerate(order)}
    ref_pos = [pos[int(k)] for k in layers]
    swaps = sum(1 for a, b in zip(ref_pos, ref_pos[1:]) if b < a and w[order[a], 0] - w[order[b], 0] > 1e-6 * abs(w[order[a], 0]))
    assert swaps == 0


@pytest.mark.parametrize('gating', [True, False])
def test_weighted_subtree_batch_equals_per_probe(gpu_device, gating):
    """weighted_subtree_ebp_batch (N probes, per-probe classifier, shared launches: 2N-stream weight pass, N-stream capture,
    J x N layerwise rounds with per-row prior tables) against the one-probe method on each probe, and -- probe 0 -- against the
    reference's golden map."""
    g = GC.golden('golden_subtree_mini')
    subj, sd, x0 = _mini('norelu')
    wb = subj.wb
    n = 3
    x = torch.cat((x0, make_images('stresnet_mini', n - 1, seed=11)), dim=0)
    xm = torch.cat((synth.unit_rows(1, 512, seed=1), synth.unit_rows(n - 1, 512, seed=31)), dim=0) / 2500
    xn = torch.cat((synth.unit_rows(1, 512, seed=2), synth.unit_rows(n - 1, 512, seed=32)), dim=0) / 2500
    single = []
    for i in range(n):
        subj.set_cls(xm[i:i + 1], xn[i:i + 1])
        single.append(wb.weighted_subtree_ebp(x[i:i + 1], 0, 1, topk=8, verbose=False, subtree_mode='norelu',
                                              do_mated_similarity_gating=gating))
    for sweep_batch in (None, 5):             # 5: several rounds, idle rows once a probe is done
        batch = wb.weighted_subtree_ebp_batch(x, xm, xn, topk=8, subtree_mode='norelu', do_mated_similarity_gating=gating,
                                              sweep_batch=sweep_batch)
        assert len(batch) == n
        for i in range(n):
            sm_b, P_b, w_b, k_b = batch[i]
            sm_s, P_s, w_s, k_s = single[i]
            assert sorted(k_b) == sorted(k_s), (i, k_b, k_s)
            assert np.allclose(sorted(w_b), sorted(w_s), rtol=1e-4)
            assert_map_close_robust(sm_b, sm_s, 'probe %d batch vs single' % i)
    if gating:
        key = 'mini/norelu/top8'
        assert [int(k) for k in batch[0][3]] == [int(k) for k in g[key + '/k_valid']]
        assert_map_close_robust(batch[0][0], g[key + '/map'], key + ' (batched)')


def test_layerwise_batch_rows_and_idle_sweeps(gpu_device):
    """xfr_layerwise_ebp with N images x J sweeps: row (j, b) equals the one-image call with the same prior; firing -1 rows stay
    zero; xfr_ebp_capture at N images equals N one-image calls."""
    subj, sd, x0 = _mini('all')
    wb = subj.wb
    eng = wb._engine(2)
    x = torch.cat((x0, make_images('stresnet_mini', 1, seed=12)), dim=0).to(gpu_device)
    st, seed = wb.net.seed_for(torch.tensor([[1.0, 0.0]]), 2)
    nf = eng.firing_count(st)
    w_, idx = eng.subtree_weights(x, st, torch.stack((seed, seed), 0))
    cap2 = eng.ebp_capture(x, st, seed.unsqueeze(0), idx)
    for b in range(2):
        cap1 = eng.ebp_capture(x[b:b + 1], st, seed[b:b + 1].unsqueeze(0), idx[:, b])
        assert np.allclose(cap2[:, b], cap1, rtol=1e-5, atol=1e-30)
    F = np.array([[3, 7], [12, -1], [33, 20]], dtype=np.int32)
    E = np.array([[idx[3, 0], idx[7, 1]], [idx[12, 0], 0], [idx[33, 0], idx[20, 1]]], dtype=np.int32)
    V = np.array([[cap2[3, 0], cap2[7, 1]], [cap2[12, 0], 0.0], [cap2[33, 0], cap2[20, 1]]], dtype=np.float32)
    maps = eng.layerwise(x, st, F, E, V).cpu().numpy()
    assert maps.shape[:2] == (3, 2) and np.all(maps[1, 1] == 0)
    for j in range(3):
        for b in range(2):
            if F[j, b] < 0:
                continue
            one = eng.layerwise(x[b:b + 1], st, [int(F[j, b])], [int(E[j, b])], [float(V[j, b])]).cpu().numpy()[0]
            if np.abs(one).max() == 0:
                assert np.abs(maps[j, b]).max() == 0
            else:
                assert_map_close(maps[j, b], one, 'row (%d, %d)' % (j, b))


@pytest.mark.parametrize('k', [5, 20, 40])
def test_layerwise_argmax_golden(gpu_device, k):
    g = GC.golden('golden_subtree_mini')
    subj, sd, x = _mini('norelu')
    got = subj.wb.layerwise_ebp(x, k_layer=k, mode='argmax', k_poschannel=0, mwp=True)
    want = g['mini/norelu/layerwise_argmax_%d' % k]
    if np.abs(want).max() == 0:
        assert np.abs(got).max() == 0
    else:
        assert_map_close_robust(got, want, 'layerwise argmax %d' % k)


def test_layerwise_elementwise_vs_oracle(gpu_device):
    from oracle import ebp_oracle as O
    subj, sd, x = _mini('all')
    ow = O.OracleWhitebox('stresnet_mini', sd, ('hooked', None), 'all')
    ow.set_triplet_classifier(synth.unit_rows(1, 512, seed=1) / 2500, synth.unit_rows(1, 512, seed=2) / 2500)
    ow.ebp(x, ow._onehot(x, 0))
    for k in (3, 12, 33):
        el = int(torch.argmax(ow.P[k]))
        want = ow.layerwise_ebp(x, k_layer=k, mode='elementwise', k_element=el, k_poschannel=0, mwp=True)
        got = subj.wb.layerwise_ebp(x, k_layer=k, mode='elementwise', k_element=el, k_poschannel=0, mwp=True)
        assert_map_close_robust(got, want, 'layerwise elementwise %d' % k)


@pytest.mark.skipif(not os.path.exists(os.path.join(GC.GOLDEN_DIR, 'golden_subtree_r101.npz')), reason='R101 subtree golden not generated')
def test_weighted_subtree_resnet101_golden(gpu_device):
    g = GC.golden('golden_subtree_r101')
    gold = GC.golden('golden_r101')
    bb, sd = make_backbone('stresnet101', seed=0, num_classes=65359)
    subj = GC.engine_subject('stresnet101', bb, 'norelu')
    subj.wb.debug_trace = False
    x_demo, x_probe, x_non, x_mate = GC.net_inputs('stresnet101')
    subj.set_cls((1.0 / 2500.0) * torch.from_numpy(gold['r101/norelu/enc_mate']), (1.0 / 2500.0) * torch.from_numpy(gold['r101/norelu/enc_nonmate']))
    key = 'r101/norelu/top32'
    smap, P_valid, w_valid, k_valid = subj.wb.weighted_subtree_ebp(x_probe, 0, 1, topk=32, verbose=False, subtree_mode='norelu')
    ref_k = [int(k) for k in g[key + '/k_valid']]
    # the same 32 layers -- except that layers whose weight TIES with the weight at the cut (several hooks see one gradient tensor,
    # hence equal weights: whitebox.py:684-696) may be chosen differently among themselves
    wk = dict(zip(ref_k, [float(v) for v in g[key + '/w_valid']]))
    wk.update(dict(zip([int(k) for k in k_valid], [float(v) for v in w_valid])))
    cut = min(wk[k] for k in ref_k)
    for k in set(k_valid) ^ set(ref_k):
        assert abs(wk[k] - cut) <= 1e-4 * cut, 'layer %d (weight %.6g) differs from the reference selection away from the cut (%.6g)' % (k, wk[k], cut)
    assert len(k_valid) == len(ref_k) == 32
    assert np.allclose(sorted(w_valid), sorted(g[key + '/w_valid']), rtol=1e-4)
    assert_map_close_robust(smap, g[key + '/map'], key, rtol=5e-3)


LAZY_ZERO_SCRIPT = r"""
import hashlib, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import golden_cases as GC
from parity_utils import make_backbone, make_images
from xfr_amd import synth
bb, sd = make_backbone('stresnet_mini', seed=3, num_classes=5)
wb = GC.engine_subject('stresnet_mini', bb, 'norelu').wb
n = 3
x = make_images('stresnet_mini', n, seed=11)
xm, xn = synth.unit_rows(n, 512, seed=31) / 2500, synth.unit_rows(n, 512, seed=32) / 2500
h = hashlib.sha256()
for sweep_batch in (None, 5):
    for sm, P, w, k in wb.weighted_subtree_ebp_batch(x, xm, xn, topk=8, subtree_mode='norelu', sweep_batch=sweep_batch):
        assert np.isfinite(np.asarray(sm)).all()
        h.update(np.ascontiguousarray(np.asarray(sm, dtype=np.float32)).tobytes())
        h.update(np.asarray(sorted(k), dtype=np.int64).tobytes())
lw = wb.layerwise_ebp(x[:1], k_poschannel=0, k_layer=7, mode='argmax', mwp=False)
h.update(np.ascontiguousarray(np.asarray(lw, dtype=np.float32)).tobytes())
print('DIGEST', h.hexdigest())
"""


def test_lazy_gradient_zeroing_against_the_eager_fill(gpu_device):
    """The layerwise / prefix sweeps zero exactly the gradient rows a launch is about to read and nobody has written (engine.hip, run_backward) instead of
    the whole region.  XFR_POISON_G=1 NaN-fills the region first, so a row the bookkeeping misses would surface in the maps; XFR_EAGER_ZERO=1 is the old
    whole-region fill.  Both are latched at first use, hence one process each: the three runs (default, poisoned, eager) must agree bit for bit."""
    import subprocess
    import sys
    root = os.path.dirname(GC.GOLDEN_DIR.rstrip('/').rsplit('/', 1)[0])
    digests = {}
    for name, extra in (('default', {}), ('poisoned', {'XFR_POISON_G': '1'}), ('eager', {'XFR_EAGER_ZERO': '1'})):
        env = {k: v for k, v in os.environ.items() if k not in ('XFR_POISON_G', 'XFR_EAGER_ZERO')}
        env.update(extra)
        out = subprocess.run([sys.executable, '-c', LAZY_ZERO_SCRIPT % (root, os.path.join(root, 'tests'))], capture_output=True, text=True, timeout=600, env=env,
                             cwd=root)
        assert out.returncode == 0, (name, out.stderr[-3000:])
        digests[name] = [ln for ln in out.stdout.splitlines() if ln.startswith('DIGEST')][-1]
    assert digests['default'] == digests['poisoned'] == digests['eager'], digests


def test_inpainting_game_workload_tool(gpu_device):
    """tools/inpainting_game_workload.py (BASELINE.json configs[4] shape) runs end to end on one GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(GC.GOLDEN_DIR.rstrip('/').rsplit('/', 1)[0])
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'inpainting_game_workload.py'), '--jobs', '3', '--mates', '2',
                          '--topk', '4', '--num-classes', '300'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    res = json.loads(line)
    assert res['jobs'] == 3 and res['jobs_per_s'] > 0


def test_workload_tool_writes_and_resumes(gpu_device, tmp_path):
    """--output-dir: every job's four maps land in the generator's npz/PNG layout; a second run recomputes nothing."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(GC.GOLDEN_DIR.rstrip('/').rsplit('/', 1)[0])
    cmd = [sys.executable, os.path.join(root, 'tools', 'inpainting_game_workload.py'), '--jobs', '2', '--mates', '2', '--topk', '4',
           '--num-classes', '300', '--output-dir', str(tmp_path)]
    written = []
    for _ in range(2):
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        written.append(json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])['maps_written_rank0'])
    assert written == [8, 0]
    files = [f for _, _, fs in os.walk(str(tmp_path)) for f in fs]
    assert sum(f.endswith('-saliency.npz') for f in files) == 8 and sum(f.endswith('-saliency-overlay.png') for f in files) == 8


def test_embeddings_sweep_tool(gpu_device):
    """tools/embeddings_sweep.py (SURVEY.md 8f row 3 shape) runs end to end."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(GC.GOLDEN_DIR.rstrip('/').rsplit('/', 1)[0])
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'embeddings_sweep.py'), '--masks', '48', '--batch', '32'],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert res['masks'] == 48 and res['images_per_s'] > 0 and -1.0 <= res['mean_similarity'] <= 1.0


@pytest.mark.parametrize('sub', ['norelu', 'affineonly_with_prior'])
def test_layerwise_contrastive_ebp_golden(gpu_device, sub):
    """layerwise_contrastive_ebp (whitebox.py:584-645; deprecated there, kept for drop-in completeness): every mode on three layers of the mini
    ResNet against the maps of the real reference (tests/golden/make_golden_lwc.py).  The prior is a CONTRAST of two MWP tensors of one layer:
    dense modes are held to the contrastive tolerance; the one-element modes (argmax, elementwise, ...) pick an element, and their whole map
    scales with that element's contrast -- compared by direction (cosine) and to 2 % in scale."""
    import warnings
    g = GC.golden('golden_lwc_mini')
    bb, sd = make_backbone('stresnet_mini', seed=3, recipe='mild', num_classes=5)
    subj = GC.engine_subject('stresnet_mini', bb, sub)
    subj.wb.debug_trace = False
    subj.set_cls(synth.unit_rows(1, 512, seed=1) / 2500, synth.unit_rows(1, 512, seed=2) / 2500)
    x = make_images('stresnet_mini', 1, seed=5)
    P0 = torch.zeros((1, 2))
    P0[0][0] = 1.0
    subj.wb.ebp(x, P0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for k in [int(v) for v in g['mini/%s/layers' % sub]]:
            kel = int(g['mini/%s/k_element_%d' % (sub, k)])
            Pk = subj.wb.P[k]
            grad = {k: Pk}
            for mode in ('copy', 'mean', 'product', 'argmax', 'argmax_product', 'percentile', 'percentile_argmax', 'elementwise'):
                want = g['mini/%s/%s_%d' % (sub, mode, k)]
                got = subj.wb.layerwise_contrastive_ebp(x, 0, 1, k_layer=k, mode=mode, percentile=80, k_element=kel, gradlayer=grad, mwp=True)
                assert got.shape == want.shape and np.isfinite(got).all(), (mode, k)
                if np.abs(want).max() == 0:
                    assert np.abs(got).max() <= 1e-12, (mode, k)
                    continue
                rel, cos = map_metrics(got, want)
                if mode in ('copy', 'mean', 'product', 'percentile'):
                    assert cos >= 0.9999 and rel <= 2e-2, (sub, mode, k, rel, cos)
                else:
                    scale = float(np.abs(got).max() / np.abs(want).max())
                    assert cos >= 0.9999 and abs(scale - 1.0) <= 2e-2, (sub, mode, k, scale, cos)

