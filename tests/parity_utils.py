"""Shared helpers of the parity tests: seeded backbones, the oracle twin, and the comparison metrics.

Tolerance (stated once, used everywhere).  The reference's own fp32-vs-fp64 self-consistency on these maps is
max|d|/max ~ 1e-4, so a GPU map is accepted when
    max|gpu - ref| / max|ref| <= 1e-3   and   cosine(gpu, ref) >= 0.99999                    (assert_map_close)
Contrastive maps are a DIFFERENCE of two normalised MWP tensors (whitebox.py:524-526).  With seeded random weights the
mate and non-mate sweeps are nearly identical, so the subtraction cancels 2-3 digits and last-bit differences of P
(1e-7) surface at 1e-4..1e-3 of the contrastive map (measured: a 1-ulp change of one classifier row moves the
reference's own map by 5e-4).  Contrastive / truncated maps are therefore accepted at
    max|d|/max <= 5e-3 (MAP_RTOL_CONTRAST)   and   cosine >= 0.99999,
while the MWP tensors they are computed from are held to the 1e-3 / 1e-4 bars above.
Two steps of the reference algorithm are DISCONTINUOUS in the activations and flip on last-bit differences between
any two fp32 implementations (MKLDNN vs MFMA summation order), moving a whole gradient element:
  * max-pool argmax near-ties (two window elements equal to ~1e-7 relative: ~10 windows per 64x112x112 image),
  * the percentile mask of truncated_contrastive_ebp (elements within rounding of the cut value).
Each flip perturbs a handful of pixels by up to a few percent of the map maximum and nothing else.  Where a case is
exposed to this, the criterion is the robust one (assert_map_close_robust): cosine >= 0.99999, at most 0.2 % of the
pixels off by more than 1e-3 of the maximum, none by more than 5e-2; the per-firing P sums (which flips preserve) are
always held to 1e-4 relative.
Where it is used (round 2): the golden cases of all backbones pass the STRICT criterion and are held to it; the robust one
remains for truncated maps (percentile mask) and for comparisons against the CPU oracle on uniform-NOISE images
(tests/test_gpu_ties.py locates the near-tie windows and shows that no pixel misses the strict tolerance outside them).
"""
import numpy as np
import torch

from xfr_amd import synth
from xfr_amd.models import lightcnn, resnet, resnet50_128

MAP_RTOL = 1e-3
MAP_RTOL_CONTRAST = 5e-3
MAP_COS = 0.99999

R50_MEAN = (131.0912, 103.8827, 91.4953)


def make_backbone(arch, seed=0, recipe='mild', num_classes=None):
    if arch == 'stresnet101':
        bb = resnet.ResNet([3, 4, 23, 3], num_classes=num_classes or 65359)
    elif arch == 'stresnet_mini':
        bb = resnet.ResNet([1, 1, 1, 1], num_classes=num_classes or 5)
    elif arch == 'resnet50_128':
        bb = resnet50_128.Resnet50_128()
    elif arch == 'lightcnn29v2':
        bb = lightcnn.LightCNN_29Layers_v2(num_classes=num_classes or 80013)
    else:
        raise ValueError(arch)
    sd = synth.synth_state_dict(bb, seed=seed, recipe=recipe)
    bb.load_state_dict(sd)
    return bb, sd


def make_images(arch, n, seed=5, smooth=True):
    f = synth.synth_smooth_images if smooth else synth.synth_images
    if arch in ('stresnet101', 'stresnet_mini'):
        return f(n, (3, 224, 224), seed=seed, mean=resnet.MEAN_RGB)
    if arch == 'resnet50_128':
        return f(n, (3, 224, 224), seed=seed, mean=R50_MEAN)
    return f(n, (1, 128, 128), seed=seed, scale255=False)


def emb_dim(arch):
    return {'stresnet101': 512, 'stresnet_mini': 512, 'resnet50_128': 128, 'lightcnn29v2': 256}[arch]


def map_metrics(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    rel = np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
    cos = float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))
    return rel, cos


def assert_map_close(got, want, what='', rtol=MAP_RTOL):
    assert np.isfinite(np.asarray(got)).all(), 'non-finite values in %s' % what
    rel, cos = map_metrics(got, want)
    assert rel <= rtol and cos >= MAP_COS, '%s: max|d|/max = %.3e (tol %.0e), cosine = %.8f (tol %.5f)' % (
        what, rel, rtol, cos, MAP_COS)
    return rel, cos


def assert_map_close_robust(got, want, what='', frac=2e-3, cap=5e-2, rtol=MAP_RTOL):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert np.isfinite(got).all(), 'non-finite values in %s' % what
    mx = max(np.abs(want).max(), 1e-300)
    d = np.abs(got - want) / mx
    rel, cos = map_metrics(got, want)
    bad = float((d > rtol).mean())
    assert cos >= MAP_COS and bad <= frac and d.max() <= cap, \
        '%s: cosine %.8f, %.3f %% pixels beyond %.0e, max|d|/max %.3e' % (what, cos, 100 * bad, rtol, d.max())
    return rel, cos


def assert_trace_close(sums, names, gsum, gnames, what='', rtol=1e-4):
    """Per-firing sum(P[i]) against the reference list (which additionally holds the image hook P[-1] the engine does
    not compute)."""
    n = len(sums)
    assert n in (len(gsum), len(gsum) - 1), '%s: %d firings vs %d in the reference' % (what, n, len(gsum))
    # names: class names, or full str(module) strings (Whitebox.P_layername since round 4, image hook included): compare the class names
    cls = lambda seq: [str(x).split('(')[0] for x in list(seq)[:n]]      # noqa: E731
    assert cls(names) == cls(gnames), '%s: firing order differs' % what
    err = np.abs(np.asarray(sums) - gsum[:n]) / np.maximum(np.abs(gsum[:n]), 1e-300)
    i = int(err.argmax())
    assert err.max() <= rtol, '%s: P-sum rel err %.3e at firing %d (%s)' % (what, err.max(), i, names[i])
