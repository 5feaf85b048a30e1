"""The golden cases of tests/golden/make_golden.py, replayable against any subject exposing the Whitebox surface
(the CPU oracle, or the HIP engine behind xfr_amd.models.whitebox)."""
import os

import numpy as np
import PIL.Image
import torch

from parity_utils import R50_MEAN, make_backbone, make_images
from xfr_amd import synth
from xfr_amd.models import lightcnn as xlightcnn
from xfr_amd.models import resnet as xresnet

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
JPEGS = ['demo_face.jpg', 'n00000001_00000117.JPEG', 'n00000002_00000100.JPEG', 'n00000001_00000384.JPEG']

_cache = {}


def golden(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False)
    return _cache[name]


def jpegs():
    g = golden('inputs_jpeg')
    return {f: g[f.replace('.', '_')] for f in JPEGS}


def net_inputs(arch):
    """(x_demo, x_probe, x_nonmate, x_mate) exactly as make_golden.py built them."""
    j = jpegs()
    if arch in ('stresnet101',):
        conv = lambda a: xresnet.convert_resnet101v4_image(a).unsqueeze(0)   # noqa: E731
    elif arch == 'resnet50_128':
        conv = lambda a: torch.from_numpy((a.astype(np.float64) - np.array(R50_MEAN)).transpose(2, 0, 1).astype(np.float32)).unsqueeze(0)  # noqa: E731
    else:
        def conv(a):
            im = PIL.Image.fromarray(a).resize((128, 128), PIL.Image.BILINEAR)
            return xlightcnn.prepare_lightCNN_image(im)
    return tuple(conv(j[f]) for f in JPEGS)


def c2_images():
    """(im_mates, im_nonmates, probe_im) of the C2 fixtures (tests/golden/make_golden_c2.py): three mates, three non-mates
    and a probe built from the bundled JPEGs, as image_loader would hand them over (float64 RGB in [0, 1], xfr/utils.py:88-90)
    or as uint8."""
    j = jpegs()
    J = [j[f] for f in JPEGS]                          # demo_face, probe, non-mate, mate
    f64 = lambda a: a.astype(float) / 255              # noqa: E731
    im_mates = [f64(J[3]), f64(J[3][:, ::-1].copy()), J[1][:, ::-1].copy()]
    im_nonmates = [f64(J[2]), f64(J[2][:, ::-1].copy()), J[0].copy()]
    return im_mates, im_nonmates, J[1].copy()


class Subject(object):
    """Uniform handle: `wb` has ebp / contrastive_ebp / truncated_contrastive_ebp; `enc` encodes; `set_cls` installs the
    triplet classifier; `trace()` returns (sums, names) of the last ebp sweep or None.

    gold_enc: when set (engine subjects), the triplet classifier rows are taken from the reference's own encodings
    stored in the golden file instead of the subject's -- contrastive EBP with near-identical mate / non-mate
    responses amplifies a 1-ulp change of a classifier row to ~5e-4 of the map, so the EBP path is compared under the
    SAME classifier and the encodings are compared separately (at 1e-4)."""

    def __init__(self, wb, enc, set_cls, trace=None, gold_enc=False):
        self.wb, self.enc, self.set_cls, self._trace, self.gold_enc = wb, enc, set_cls, trace, gold_enc

    def encodings(self, gold, key_m, key_n, x_mate, x_non):
        em, en = self.enc(x_mate).detach().cpu(), self.enc(x_non).detach().cpu()
        if self.gold_enc:
            gm, gn = torch.from_numpy(gold[key_m]), torch.from_numpy(gold[key_n])
            for a, b in ((em, gm), (en, gn)):
                assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()), 'encode() differs from the reference'
            return gm, gn
        return em, en

    def trace(self):
        return self._trace() if self._trace else None


def oracle_subject(arch, sd, mode, num_classes=None):
    from oracle import ebp_oracle as O
    ow = O.OracleWhitebox(arch, sd, ('hooked', None), mode)

    def trace():
        return (np.array([float(p.double().sum()) for p in ow.P]), list(ow.P_layername))
    return Subject(ow, ow.encode, ow.set_triplet_classifier, trace)


class _Replicated(object):
    """Whitebox proxy for the golden replay: contrastive calls run on FOUR copies of the probe and return row 0, so that the batch is a
    multiple of four and the sweep takes the engine's lean schedule (xfr_engine_set_lean) -- the golden vectors then pin that schedule to the
    reference too, not only the literal one a batch of one runs.  Everything else passes through."""

    def __init__(self, wb):
        object.__setattr__(self, '_wb', wb)

    def __getattr__(self, name):
        return getattr(self._wb, name)

    def __setattr__(self, name, value):
        setattr(self._wb, name, value)

    def contrastive_ebp(self, x, kp, kn):
        return self._wb.contrastive_ebp(x.repeat(4, 1, 1, 1), kp, kn)[0]

    def truncated_contrastive_ebp(self, x, kp, kn, percentile=20):
        return self._wb.truncated_contrastive_ebp(x.repeat(4, 1, 1, 1), kp, kn, percentile)[0]


def engine_subject(arch, bb, mode, device='cuda:0', replicate=False):
    from xfr_amd.models import whitebox as WB
    bb.to(device)
    if arch == 'resnet50_128':
        wbn = WB.Whitebox_resnet50_128(bb)
    elif arch == 'lightcnn29v2':
        wbn = WB.WhiteboxLightCNN(bb)
    else:
        wbn = WB.WhiteboxSTResnet(bb)
    wb = WB.Whitebox(wbn, ebp_subtree_mode=mode)
    wb.debug_trace = True

    def trace():
        if not getattr(wb, 'P_layername', None):
            return None
        return (np.asarray(wb.P_trace)[:, 0], list(wb.P_layername))
    return Subject(_Replicated(wb) if replicate else wb, wbn.encode, wbn.set_triplet_classifier, trace, gold_enc=True)


# --------------------------------------------------------------------------------------------------------------
# case tables: key -> function(subject, ctx) -> result map.  ctx carries inputs.
# --------------------------------------------------------------------------------------------------------------
def mini_cases(recipe, mode):
    x = make_images('stresnet_mini', 1, seed=5)
    xm = synth.unit_rows(1, 512, seed=1) / 2500
    xn = synth.unit_rows(1, 512, seed=2) / 2500
    Pn = torch.zeros(1, 5)
    Pn[0, 2] = 1
    P2 = torch.zeros(1, 2)
    P2[0, 1] = 1
    pre = 'mini/%s/%s/' % (recipe, mode)
    cases = [(pre + 'hooked/ebp', lambda s: s.wb.ebp(x, Pn, mwp=True)),
             ('__set__', lambda s: s.set_cls(xm, xn)),
             (pre + 'triplet/ebp', lambda s: s.wb.ebp(x, P2, mwp=True))]
    if recipe == 'mild':
        cases += [(pre + 'triplet/contrastive', lambda s: s.wb.contrastive_ebp(x, 0, 1)),
                  (pre + 'triplet/truncated', lambda s: s.wb.truncated_contrastive_ebp(x, 0, 1, 20))]
    return cases


def r101_cases(mode, which=None):
    x_demo, x_probe, x_non, x_mate = net_inputs('stresnet101')
    NC = 65359
    P = torch.zeros((1, NC))
    P[0][0] = 1.0
    P2 = torch.zeros((1, 2))
    P2[0][0] = 1.0
    pre = 'r101/%s/' % mode
    cases = []
    if mode == 'affineonly_with_prior':
        cases += [(pre + 'hooked/ebp', lambda s: s.wb.ebp(x_demo, P)),
                  (pre + 'hooked/contrastive', lambda s: s.wb.contrastive_ebp(x_demo, 0, 100)),
                  (pre + 'hooked/truncated', lambda s: s.wb.truncated_contrastive_ebp(x_demo, 0, 100, 20))]

    gold = golden('golden_r101')

    def set_real(s):   # demo/test_whitebox.py:129 multiplies by the reciprocal
        em, en = s.encodings(gold, pre + 'enc_mate', pre + 'enc_nonmate', x_mate, x_non)
        s.set_cls((1.0 / 2500.0) * em, (1.0 / 2500.0) * en)
    imgs = synth.synth_images(3, (3, 224, 224), seed=1234, mean=xresnet.MEAN_RGB)

    def set_synth(s):
        s.set_cls(synth.unit_rows(1, 512, seed=1) / 2500, synth.unit_rows(1, 512, seed=2) / 2500)
    cases += [('__set__', set_real),
              (pre + 'triplet/ebp', lambda s: s.wb.ebp(x_probe, P2)),
              (pre + 'triplet/contrastive', lambda s: s.wb.contrastive_ebp(x_probe, 0, 1)),
              (pre + 'triplet/truncated', lambda s: s.wb.truncated_contrastive_ebp(x_probe, 0, 1, 20)),
              ('__set__', set_synth),
              (pre + 'synthetic/contrastive', lambda s: s.wb.contrastive_ebp(imgs[2:3], 0, 1))]
    return _filter(cases, which)


def r50_cases(mode, which=None):
    x_demo, x_probe, x_non, x_mate = net_inputs('resnet50_128')
    P2 = torch.zeros((1, 2))
    P2[0][0] = 1.0
    pre = 'r50/%s/' % mode
    gold = golden('golden_r50')

    def set_real(s):
        em, en = s.encodings(gold, pre + 'enc_mate', pre + 'enc_nonmate', x_mate, x_non)
        s.set_cls(em / 2500.0, en / 2500.0)
    cases = [('__set__', set_real),
             (pre + 'triplet/ebp', lambda s: s.wb.ebp(x_demo, P2, mwp=False)),
             (pre + 'triplet/contrastive', lambda s: s.wb.contrastive_ebp(x_probe, 0, 1)),
             (pre + 'triplet/truncated', lambda s: s.wb.truncated_contrastive_ebp(x_probe, 0, 1, 20))]
    return _filter(cases, which)


def lcnn_cases(mode, which=None):
    g = golden('golden_lcnn')
    x_demo, x_probe, x_non, x_mate = (torch.from_numpy(g['lcnn/' + k]) for k in ('x_demo', 'x_probe', 'x_non', 'x_mate'))
    NCL = 80013
    P = torch.zeros((1, NCL))
    P[0][0] = 1.0
    pre = 'lcnn/%s/' % mode

    def set_real(s):
        em, en = s.encodings(g, pre + 'enc_mate', pre + 'enc_nonmate', x_mate, x_non)
        s.set_cls(em / 2500.0, en / 2500.0)
    cases = [(pre + 'hooked/ebp', lambda s: s.wb.ebp(x_demo, P, mwp=False))]
    if mode != 'affineonly':
        cases += [('__set__', set_real),
                  (pre + 'triplet/contrastive', lambda s: s.wb.contrastive_ebp(x_probe, 0, 1)),
                  (pre + 'triplet/truncated', lambda s: s.wb.truncated_contrastive_ebp(x_probe, 0, 1, 20))]
    return _filter(cases, which)


def synth_cases(arch, tag, mode):
    """tests/golden/make_golden_synth.py: a smooth synthetic probe against two independent random classifier rows -- a WELL-CONDITIONED contrast."""
    from parity_utils import emb_dim
    x = make_images(arch, 1, seed=77, smooth=True)
    D = emb_dim(arch)
    pre = '%s/%s/synthetic/' % (tag, mode)
    return [('__set__', lambda s: s.set_cls(synth.unit_rows(1, D, seed=1) / 2500, synth.unit_rows(1, D, seed=2) / 2500)),
            (pre + 'contrastive', lambda s: s.wb.contrastive_ebp(x, 0, 1)),
            (pre + 'truncated', lambda s: s.wb.truncated_contrastive_ebp(x, 0, 1, 20))]


def _filter(cases, which):
    if which is None:
        return cases
    return [c for c in cases if c[0] == '__set__' or any(c[0].endswith(w) for w in which)]


def replay(subject, cases, gold, check):
    """Run `cases` in order on `subject`; for every non-setup case call check(key, result, trace, gold)."""
    for key, fn in cases:
        if key == '__set__':
            fn(subject)
            continue
        res = fn(subject)
        check(key, res, subject.trace(), gold)
