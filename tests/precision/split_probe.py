"""Precision study (CPU, test infrastructure, not collected by pytest): what would a bf16-split MFMA GEMM do to the maps?

SURVEY.md section 7 lists "bf16x3 split emulation" as the one lever past the fp32 MFMA peak (157 TFLOP/s against 2.5 PFLOP/s bf16 dense).
Before anybody builds such a kernel, this script answers the gating question on the CPU: it replays golden cases through the ORACLE with
every convolution / linear layer (true pass, positive pass and the W+ backward) computed as a sum of products of bf16 PIECES of both
operands, accumulated in fp32 -- exactly what a split kernel would feed v_mfma_f32_32x32x16_bf16 (a bf16 x bf16 product is exact in fp32) --
and reports each map's distance from the reference's golden map next to the plain-fp32 oracle's.

    pieces 2, terms i+j <= 1  -> "bf16x3"  (3 MFMAs per fp32 MFMA's work, product error ~2^-16)
    pieces 3, terms i+j <= 2  -> "bf16x6"  (6 MFMAs, ~2^-23)
    pieces 3, all 9 terms     -> "bf16x9"

    python tests/precision/split_probe.py [--arch r101|r50|lcnn] [--threads 8]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import golden_cases as GC  # noqa: E402
from parity_utils import make_backbone, map_metrics  # noqa: E402
from oracle import ebp_oracle as O  # noqa: E402

REAL_CONV2D, REAL_LINEAR = F.conv2d, F.linear


def pieces(t, n):
    out, r = [], t
    for _ in range(n):
        p = r.bfloat16().float()
        out.append(p)
        r = r - p          # exact: p is r rounded to 8 significant bits
    return out


def pairs(n, max_order):
    pr = [(i, j) for i in range(n) for j in range(n) if i + j <= max_order]
    return sorted(pr, key=lambda ij: -(ij[0] + ij[1]))      # smallest terms first


class SplitConv(torch.autograd.Function):
    cfg = (3, 2)

    @staticmethod
    def forward(ctx, x, w, stride, pad):
        n, mo = SplitConv.cfg
        ctx.save_for_backward(x, w)
        ctx.sp = (stride, pad)
        xs, ws = pieces(x, n), pieces(w, n)
        out = None
        for i, j in pairs(n, mo):
            t = REAL_CONV2D(xs[i], ws[j], None, stride=stride, padding=pad)
            out = t if out is None else out + t
        return out

    @staticmethod
    def backward(ctx, g):
        n, mo = SplitConv.cfg
        x, w = ctx.saved_tensors
        stride, pad = ctx.sp
        gs, ws = pieces(g.contiguous(), n), pieces(w, n)
        out = None
        for i, j in pairs(n, mo):
            t = torch.nn.grad.conv2d_input(x.shape, ws[j], gs[i], stride=stride, padding=pad)
            out = t if out is None else out + t
        return out, None, None, None


def split_conv2d(x, w, b=None, stride=1, padding=0):
    out = SplitConv.apply(x, w, stride, padding)
    return out if b is None else out + b.reshape(1, -1, 1, 1)


def split_linear(x, w, b=None):
    out = SplitConv.apply(x.reshape(x.shape[0], x.shape[1], 1, 1), w.reshape(w.shape[0], w.shape[1], 1, 1), 1, 0).reshape(x.shape[0], w.shape[0])
    return out if b is None else out + b


class _F(object):
    """The oracle's `F` with the two GEMM-shaped calls swapped."""

    def __init__(self, split):
        self.split = split

    def __getattr__(self, k):
        if self.split and k == 'conv2d':
            return split_conv2d
        if self.split and k == 'linear':
            return split_linear
        return getattr(F, k)


def run(arch, cfgname):
    cfgs = {'fp32': None, 'bf16x3': (2, 1), 'bf16x6': (3, 2), 'bf16x9': (3, 4)}
    c = cfgs[cfgname]
    O.F = _F(c is not None)
    if c:
        SplitConv.cfg = c
    rows = []

    def check(key, res, trace, gold):
        rel, cos = map_metrics(res, gold[key + '/map'])
        sums, _ = trace
        gsum = gold[key + '/trace']
        terr = float((np.abs(sums - gsum) / np.maximum(np.abs(gsum), 1e-300)).max())
        rows.append((key, rel, cos, terr))
    try:
        if arch == 'r101':
            gold = GC.golden('golden_r101')
            bb, sd = make_backbone('stresnet101', seed=0, num_classes=65359)
            mode = 'affineonly_with_prior'
            cases = GC.r101_cases(mode, which=['hooked/ebp', 'triplet/ebp', 'triplet/contrastive', 'synthetic/contrastive', 'triplet/truncated'])
            GC.replay(GC.oracle_subject('stresnet101', sd, mode), cases, gold, check)
        elif arch == 'r50':
            bb, sd = make_backbone('resnet50_128', seed=0)
            GC.replay(GC.oracle_subject('resnet50_128', sd, 'norelu'), GC.r50_cases('norelu'), GC.golden('golden_r50'), check)
            GC.replay(GC.oracle_subject('resnet50_128', sd, 'norelu'), GC.synth_cases('resnet50_128', 'r50', 'norelu'), GC.golden('golden_synth'), check)
        else:
            bb, sd = make_backbone('lightcnn29v2', seed=0, num_classes=80013)
            mode = 'affineonly_with_prior'
            GC.replay(GC.oracle_subject('lightcnn29v2', sd, mode), GC.lcnn_cases(mode), GC.golden('golden_lcnn'), check)
    finally:
        O.F = F
    return rows


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='r101')
    ap.add_argument('--threads', type=int, default=8)
    ap.add_argument('--cfgs', default='fp32,bf16x3,bf16x6,bf16x9')
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    print('# %s: max|d|/max and cosine against the reference\'s golden map; worst relative error of a per-firing P sum' % a.arch)
    for name in a.cfgs.split(','):
        for key, rel, cos, terr in run(a.arch, name):
            print('%-7s %-46s rel %.3e  1-cos %.2e  P-sum err %.2e' % (name, key, rel, 1 - cos, terr), flush=True)
