"""Shared helpers of the parity tests: seeded backbones, the oracle twin, and the comparison metrics.

Tolerance (stated once, used everywhere): the reference's own fp32-vs-fp64 self-consistency on these maps is
max|d|/max ~ 1e-4, so a GPU map is accepted when
    max|gpu - oracle| / max|oracle| <= 1e-3   and   cosine(gpu, oracle) >= 0.99999.
"""
import numpy as np
import torch

from xfr_amd import synth
from xfr_amd.models import lightcnn, resnet, resnet50_128

MAP_RTOL = 1e-3
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


def assert_map_close(got, want, what=''):
    assert np.isfinite(np.asarray(got)).all(), 'non-finite values in %s' % what
    rel, cos = map_metrics(got, want)
    assert rel <= MAP_RTOL and cos >= MAP_COS, '%s: max|d|/max = %.3e (tol %.0e), cosine = %.8f (tol %.5f)' % (
        what, rel, MAP_RTOL, cos, MAP_COS)
    return rel, cos
