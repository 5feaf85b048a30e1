"""Where the robust map criterion (parity_utils.assert_map_close_robust) is used instead of the strict one, WHY: this test locates
every pixel of the channel-pooled P[-2] that misses the strict 1e-3 tolerance against the reference's golden vector and shows it
lies in the footprint of a max-pool window whose two largest inputs agree to ~1e-6 relative -- the windows whose argmax two
correct fp32 implementations (MKLDNN on the CPU, MFMA here) may legitimately resolve differently, moving one gradient element
between two neighbouring positions (whitebox.py:410 'MaxPool' hooks sit on exactly that tensor).  Run with -m gpu."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_cases as GC
from parity_utils import make_backbone, map_metrics
from xfr_amd import synth
from xfr_amd.program import OpKind

pytestmark = pytest.mark.gpu
TIE_REL = 2e-6          # two window inputs this close are a near-tie: fp32 convolutions differ by about this much between implementations


def _near_tie_footprint(act, k, stride, pad, ceil_mode, out_hw):
    """act: 1 x C x H x W (>= 0, the max-pool input).  Boolean H x W mask: positions that belong to a window (any channel) whose
    two largest values are positive and within TIE_REL of each other; also the number of such windows."""
    _, C, H, W = act.shape
    OH, OW = out_hw
    need_h = (OH - 1) * stride + k - H - pad
    need_w = (OW - 1) * stride + k - W - pad
    xp = F.pad(act, (pad, max(need_w, 0), pad, max(need_h, 0)), value=-1.0)
    cols = F.unfold(xp, kernel_size=k, stride=stride).reshape(C, k * k, -1)[:, :, :OH * OW]          # C x kk x L
    top2 = cols.topk(2, dim=1).values
    tie = (top2[:, 0] > 0) & ((top2[:, 0] - top2[:, 1]) <= TIE_REL * top2[:, 0])                     # C x L
    # positions of the tied maxima: every window element within TIE_REL of the window maximum
    hot = (cols >= (top2[:, 0:1] * (1 - TIE_REL))) & tie[:, None, :]                                # C x kk x L
    back = F.fold(hot.float().reshape(1, C * k * k, -1), output_size=xp.shape[2:], kernel_size=k, stride=stride)
    mask = (back[0].sum(dim=0) > 0)[pad:pad + H, pad:pad + W]
    return mask.numpy(), int(tie.sum())


@pytest.mark.parametrize('arch,recipe,mode,key', [
    ('stresnet_mini', 'mild', 'norelu', 'mini/mild/norelu/hooked/ebp'),
    ('stresnet_mini', 'harsh', 'affineonly_with_prior', 'mini/harsh/affineonly_with_prior/hooked/ebp'),
    ('stresnet_mini', 'mild', 'all', 'mini/mild/all/triplet/ebp'),
    ('resnet50_128', 'mild', 'norelu', 'r50/norelu/triplet/ebp'),
    ('resnet50_128', 'mild', 'affineonly_with_prior', 'r50/affineonly_with_prior/triplet/ebp'),
])
def test_pixels_beyond_strict_tolerance_sit_on_maxpool_near_ties(gpu_device, arch, recipe, mode, key):
    gold = GC.golden('golden_mini' if arch == 'stresnet_mini' else 'golden_r50')
    bb, sd = make_backbone(arch, seed=3 if arch == 'stresnet_mini' else 0, recipe=recipe, num_classes=5 if arch == 'stresnet_mini' else None)
    subj = GC.engine_subject(arch, bb, mode)
    subj.wb.debug_trace = False
    want = gold[key + '/pooled'].astype(np.float64)
    got = None
    cases = GC.mini_cases(recipe, mode) if arch == 'stresnet_mini' else GC.r50_cases(mode)
    for ckey, fn in cases:                                    # replay up to the case (installs the same classifier)
        if ckey == '__set__':
            fn(subj)
        elif ckey == key:
            fn(subj)
            got = np.squeeze(np.sum(subj.wb.P[-2].cpu().numpy(), axis=1)).astype(np.float64)
            x = subj.wb.P._x
            break
    assert got is not None and got.shape == want.shape
    # the max-pool input of the engine's own forward
    prog = subj.wb.net._program
    mp = [o for o in prog.ops if o.kind == OpKind.MAXPOOL]
    assert len(mp) == 1
    eng = subj.wb._engine(1)
    act = eng.forward(x, mp[0].in0).cpu()
    assert float(act.min()) >= 0.0                             # it is the in-place ReLU's output
    _, oh, ow = eng.tensor_shape(mp[0].out)
    mask, n_ties = _near_tie_footprint(act, mp[0].kh, mp[0].stride, mp[0].pad, bool(mp[0].ceil_mode), (oh, ow))
    d = np.abs(got - want) / max(np.abs(want).max(), 1e-300)
    bad = d > 1e-3
    rel, cos = map_metrics(got, want)
    print('%s: %d near-tie windows (footprint %d px of %d), %d px beyond 1e-3 (max %.2e), cosine %.8f'
          % (key, n_ties, int(mask.sum()), mask.size, int(bad.sum()), d.max(), cos))
    assert mask.shape == bad.shape
    outside = bad & ~mask
    assert not outside.any(), '%d pixels miss the strict tolerance away from any max-pool near-tie (max %.2e)' % (int(outside.sum()), d[outside].max())
    assert bad.sum() <= mask.sum() and cos >= 0.99999
