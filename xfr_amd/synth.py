"""Seeded synthetic parameters and inputs.

The reference ships no weights (every .pth under /root/reference/models is a git-LFS pointer), so golden
vectors, parity tests, smoke() and bench.py all use parameters drawn from a seeded CPU torch.Generator.
The same image runs here and on the GPU box, so the same seed gives the same bits in both places; the
golden fixtures store a checksum of the parameters they were produced with to detect drift.

Recipes: convolutions follow the reference initialiser (resnet.py:191-198, weights AND
biases ~ N(0, sqrt(2/(k*k*Cout)))); BatchNorm statistics are randomised so that negative gammas, non-zero
betas/means occur -- 'mild' keeps every subtree mode finite on all backbones, 'harsh' provokes the
x == 0 < a corner (p / eps blow-ups) and is only used where the reference itself stays finite.
"""
import hashlib
import math
from collections import OrderedDict

import numpy as np
import torch

_BN = {
    'mild': dict(g=(1.0, 0.2), b=0.1, m=0.1, v=(0.8, 1.2)),
    'harsh': dict(g=(0.5, 0.5), b=0.3, m=0.3, v=(0.5, 1.5)),
}


def synth_state_dict(backbone, seed=0, recipe='mild'):
    r = _BN[recipe]
    g = torch.Generator().manual_seed(int(seed))
    sd = OrderedDict()
    for name, shape, kind in backbone.param_specs():
        if kind == 'conv_w':
            std = math.sqrt(2.0 / (shape[2] * shape[3] * shape[0]))
            t = torch.randn(shape, generator=g) * std
        elif kind == 'conv_b':
            t = torch.randn(shape, generator=g) * 0.05
        elif kind == 'bn_w':
            t = r['g'][0] + r['g'][1] * torch.randn(shape, generator=g)
        elif kind == 'bn_b':
            t = r['b'] * torch.randn(shape, generator=g)
        elif kind == 'bn_mean':
            t = r['m'] * torch.randn(shape, generator=g)
        elif kind == 'bn_var':
            t = r['v'][0] + (r['v'][1] - r['v'][0]) * torch.rand(shape, generator=g)
        elif kind == 'bn_nbt':
            t = torch.zeros((), dtype=torch.int64)
        elif kind == 'fc_w':
            t = torch.randn(shape, generator=g) * math.sqrt(1.0 / shape[1])
        elif kind == 'fc_b':
            t = torch.randn(shape, generator=g) * 0.05
        else:
            raise ValueError(kind)
        sd[name] = t
    return sd


def state_checksum(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            h.update(k.encode())
            h.update(v.detach().cpu().float().contiguous().numpy().tobytes())
    return h.hexdigest()[:16]


def synth_images(n, shape, seed=1234, mean=None, scale255=True):
    """uint8-valued images ~U[0,255] (as float32, NCHW), minus the per-channel mean (resnet.py:23,
    whitebox.py:239); scale255=False gives U[0,1) gray images for Light-CNN (lightcnn.py:24)."""
    g = torch.Generator().manual_seed(int(seed))
    c, h, w = shape
    if scale255:
        x = torch.randint(0, 256, (n, c, h, w), generator=g).float()
        if mean is not None:
            x = x - torch.tensor(mean, dtype=torch.float32).view(1, c, 1, 1)
    else:
        x = torch.rand((n, c, h, w), generator=g)
    return x


def synth_smooth_images(n, shape, seed=1234, mean=None, scale255=True):
    """Low-frequency 'face-like' blobs (sum of a few random Gaussians) plus 15 % pixel noise, so saliency maps have
    structure but no two receptive fields are identical (exact plateaus make max-pool argmax a coin toss between
    implementations that differ in the last bit)."""
    g = torch.Generator().manual_seed(int(seed))
    c, h, w = shape
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32),
                            indexing='ij')
    out = torch.zeros((n, c, h, w))
    for i in range(n):
        for ch in range(c):
            img = torch.zeros((h, w))
            for _ in range(6):
                cy, cx = torch.rand(2, generator=g) * torch.tensor([h, w], dtype=torch.float32)
                s = 8 + 40 * torch.rand(1, generator=g)
                amp = torch.rand(1, generator=g)
                img += amp * torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
            img = 0.85 * img / img.max() + 0.15 * torch.rand((h, w), generator=g)
            out[i, ch] = img
    if scale255:
        out = torch.floor(out * 255.0)
        if mean is not None:
            out = out - torch.tensor(mean, dtype=torch.float32).view(1, c, 1, 1)
    return out


def bench_images(batch, shape, seed=1234, mean=None):
    """The 3*batch images of one bench.py step: [mates | non-mates | probes].  Smooth seeded images (see synth_smooth_images):
    the maps the benchmark produces are then meaningful saliency maps, and sample 0 can be checked against the map the
    reference computes for the same triplet (tests/golden/golden_bench.npz).  Only the first 3 images of each third are
    generated one by one; the rest are seeded mixtures of those and of per-image noise -- distinct inputs at a fraction of the
    generation cost (the benchmark's time does not depend on the pixel values)."""
    base = synth_smooth_images(9, shape, seed=seed, mean=None)            # values in [0, 255]
    g = torch.Generator().manual_seed(int(seed) + 7)
    out = torch.empty((3 * batch,) + tuple(shape))
    for third in range(3):
        b3 = base[3 * third:3 * third + 3]
        for i in range(batch):
            if i < 3:
                img = b3[i]
            else:
                w = torch.rand(3, generator=g)
                w = w / w.sum()
                img = torch.floor((w.view(3, 1, 1, 1) * b3).sum(dim=0) * 0.9 + 25.5 * torch.rand(tuple(shape), generator=g))
            out[third * batch + i] = img
    if mean is not None:
        out = out - torch.tensor(mean, dtype=torch.float32).view(1, shape[0], 1, 1)
    return out


def unit_rows(n, d, seed=7):
    g = torch.Generator().manual_seed(int(seed))
    x = torch.randn((n, d), generator=g)
    return x / x.norm(dim=1, keepdim=True)
