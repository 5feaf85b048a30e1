"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.

Hook-free, autograd-graph-free restatement of the excitation-backprop (EBP) saliency
algorithm of stresearch/xfr, `python/xfr/models/whitebox.py` (all file:line citations are
relative to /root/reference).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module; `xfr_amd/` never does.

Parity status: PINNED.  The reference ships no numeric tests and no weights (all .pth files
are git-LFS pointers), so the pins are golden vectors captured by importing the *real*
reference in the build container with seeded synthetic weights
(`tests/golden/make_golden.py`, committed next to the vectors it produced).  This oracle is
checked against every one of them in `tests/test_oracle_golden.py`, and -- when
/root/reference is present -- P-tensor by P-tensor against the live reference in
`tests/test_oracle_vs_reference.py`.

What the reference does (whitebox.py:482-504 `ebp`), restated without hooks
---------------------------------------------------------------------------
The reference registers a pre-forward and a forward hook on every leaf nn.Module
(whitebox.py:303, `_layer_visitor` :34-56) and runs THREE forwards plus one autograd backward:

  1. mode 'activation'          (:490-491, hook :353-360):  A[k][j] = relu(true input j of call k)
  2. mode 'positive_activation' (:492-493, hook :315-330):  weights with a `.weight` attribute are
     replaced by relu(weight) (Conv, Linear *and* BatchNorm gamma, :317-320; biases only when
     with_bias, :321-324); X[k][j] = relu(input j reaching call k in this pass) (:327); the input is
     then OVERRIDDEN with A[k] (:328-330) -- so positive values only ever travel through one
     hooked module (plus un-hooked glue such as view / F.normalize / torch.add / torch.max).
  3. mode 'ebp' (:496-498, hook :365-433): true forward; after each module's forward its weight
     storage is overwritten with relu(weight) (:371-377) so that autograd's VJP uses W+; a tensor
     hook `_backward_ebp` is registered on every module input (:379-432).
  4. `Xn.backward(Pn)` (:498): each tensor hook computes zh=relu(z); p=a*zh; P.append(p); and returns
     p/(x+eps) | zh | None depending on `ebp_subtree_mode` and the module type (:396-430).

Hook placement facts (measured against the live reference):
  * a hook lives on the tensor that is input j of call k -- except for an in-place ReLU, whose
    hook lands on its own (post-ReLU) output because the forward hook runs after the in-place op;
  * several hooks on one tensor fire in registration (= call) order, each seeing the previous
    one's return value; they fire once, on the fully accumulated gradient of that tensor;
  * autograd executes nodes in strictly descending creation order, so the P list is ordered by
    descending producer call index, then registration order;
  * the closure in :379-432 late-binds (a, x): for the two-input `Add` module BOTH hooks use the
    (a, x) of the LAST input (the residual);
  * `set_triplet_classifier` (:93-96, :121-124, :218-220) creates the 2-way classifier after the
    hooks were installed, so it is un-hooked: its VJP uses its true signed weights.

This file implements exactly that as a tape: every hooked module call and every glue op of the
three supported backbones is recorded while the true forward runs; the positive pass and the
backward sweep are then evaluated from the tape with explicit per-op VJPs (local torch.autograd on
one op at a time -- the same ATen kernels the reference dispatches to, so results agree to the
last bit on the same machine).
"""
import math

import numpy as np
import scipy.ndimage
import torch
import torch.nn.functional as F

SUBTREE_MODES = ('affineonly', 'affineonly_with_prior', 'norelu', 'all')


def _is_affine(name):
    # whitebox.py:399 / :409 -- substring test on str(module)
    return ('Conv' in name) or ('Linear' in name) or ('AvgPool' in name) or ('BatchNorm' in name)


class _Call(object):
    __slots__ = ('name', 'ins', 'out', 'hooked', 'fn', 'inplace')

    def __init__(self, name, ins, out, hooked, fn, inplace=False):
        self.name = name      # class name of the reference module ('Conv2d', 'ReLU', ...) or glue tag
        self.ins = ins        # tensor ids
        self.out = out        # tensor id
        self.hooked = hooked  # True: leaf nn.Module call (hooks apply); False: glue
        self.fn = fn          # fn(list_of_inputs, positive: bool) -> output
        self.inplace = inplace


class Tape(object):
    """Records one forward of a backbone as hooked module calls + glue, evaluating true values eagerly."""

    def __init__(self, params, with_bias=False):
        self.p = params
        self.with_bias = bool(with_bias)
        self.calls = []
        self.T = []  # true value per tensor id

    # -- plumbing ---------------------------------------------------------------------------
    def input(self, x):
        self.T.append(x.detach().clone().float())
        return 0

    def _record(self, name, ins, hooked, fn, inplace=False):
        with torch.no_grad():
            out_val = fn([self.T[i] for i in ins], False)
        self.T.append(out_val)
        out = len(self.T) - 1
        self.calls.append(_Call(name, list(ins), out, hooked, fn, inplace))
        return out

    def _w(self, name, positive):
        w = self.p[name]
        return F.relu(w) if positive else w  # whitebox.py:319 pos_weight = relu(orig_weight)

    def _b(self, name, positive):
        if name is None or name not in self.p:
            return None
        b = self.p[name]
        return F.relu(b) if (positive and self.with_bias) else b  # whitebox.py:321-324

    # -- hooked leaf modules ----------------------------------------------------------------
    def conv(self, x, prefix, stride=1, pad=0):
        bname = prefix + '.bias'

        def fn(ins, positive):
            return F.conv2d(ins[0], self._w(prefix + '.weight', positive), self._b(bname, positive),
                            stride=stride, padding=pad)
        return self._record('Conv2d', [x], True, fn)

    def linear(self, x, prefix):
        bname = prefix + '.bias'

        def fn(ins, positive):
            return F.linear(ins[0], self._w(prefix + '.weight', positive), self._b(bname, positive))
        return self._record('Linear', [x], True, fn)

    def batchnorm(self, x, prefix, eps=1e-5):
        def fn(ins, positive):
            return F.batch_norm(ins[0], self.p[prefix + '.running_mean'], self.p[prefix + '.running_var'],
                                self._w(prefix + '.weight', positive), self._b(prefix + '.bias', positive),
                                False, 0.1, eps)
        return self._record('BatchNorm2d', [x], True, fn)

    def relu_(self, x):
        # nn.ReLU(inplace=True): resnet.py:124, resnet50_128.py:15
        return self._record('ReLU', [x], True, lambda ins, positive: F.relu(ins[0]), inplace=True)

    def maxpool(self, x, k, s, p=0, ceil_mode=False):
        return self._record('MaxPool2d', [x], True,
                            lambda ins, positive: F.max_pool2d(ins[0], k, s, p, 1, ceil_mode))

    def avgpool(self, x, k, s):
        return self._record('AvgPool2d', [x], True, lambda ins, positive: F.avg_pool2d(ins[0], k, s))

    def add(self, a, b):
        # resnet.py:104-108 / lightcnn.py:33-37 `Add` module
        return self._record('Add', [a, b], True, lambda ins, positive: ins[0] + ins[1])

    def concat_channels(self, x, channels):
        # resnet.py:152-157: cat((x, zeros.repeat(1, channels, 1, 1)), dim=1)
        def fn(ins, positive):
            z = torch.zeros_like(ins[0]).repeat(1, channels, 1, 1)
            return torch.cat((ins[0], z), dim=1)
        return self._record('ConcatChannels', [x], True, fn)

    def multiply(self, x, n):
        return self._record('Multiply', [x], True, lambda ins, positive: ins[0] * n)  # resnet.py:160-165

    def split(self, x):
        # lightcnn.py:39-45 Split module; its outputs only feed the glue torch.max (lightcnn.py:62), so the
        # tuple is represented by the un-split tensor and the max is the glue op `g_max_halves`.
        return self._record('Split', [x], True, lambda ins, positive: ins[0])

    # -- glue (not nn.Modules in the reference => no hooks; evaluated on whatever reaches them) ----
    def g_flatten(self, x):
        return self._record('view', [x], False, lambda ins, positive: ins[0].reshape(ins[0].shape[0], -1))

    def g_normalize(self, x):
        return self._record('normalize', [x], False, lambda ins, positive: F.normalize(ins[0], p=2, dim=1))

    def g_add(self, a, b):
        return self._record('add', [a, b], False, lambda ins, positive: ins[0] + ins[1])

    def g_max_halves(self, x):
        def fn(ins, positive):
            a, b = torch.split(ins[0], ins[0].shape[1] // 2, 1)
            return torch.max(a, b)
        return self._record('max', [x], False, fn)

    def g_linear(self, x, weight):
        # un-hooked triplet classifier (true signed weights), whitebox.py:93-96
        w = weight.detach().clone().float()
        return self._record('classifier', [x], False, lambda ins, positive: F.linear(ins[0], w))

    # -- the three passes ---------------------------------------------------------------------
    def positive_pass(self):
        """Pv[t]: value of tensor t in the reference's 'positive_activation' forward (whitebox.py:315-330)."""
        Pv = [None] * len(self.T)
        Pv[0] = self.T[0]
        with torch.no_grad():
            for c in self.calls:
                if c.hooked:
                    Pv[c.out] = c.fn([F.relu(self.T[i]) for i in c.ins], True)  # input overridden by A
                else:
                    Pv[c.out] = c.fn([Pv[i] for i in c.ins], False)
        return Pv

    def hooks_by_tensor(self):
        hooks = {}
        for k, c in enumerate(self.calls):
            if not c.hooked:
                continue
            for j, t in enumerate(c.ins):
                ht = c.out if c.inplace else t
                hooks.setdefault(ht, []).append((k, j))
        return hooks

    def backward(self, seed_tensor, seed, mode='affineonly_with_prior', eps=1e-16, priors=None):
        """The autograd sweep of whitebox.py:498 with the tensor hooks of :381-430 applied explicitly.

        priors: optional dict {firing index -> tensor} (P_prior list of :390-392, used by layerwise_ebp).
        Returns (P list, P_layername list) in the reference's firing order.
        """
        if mode not in SUBTREE_MODES:
            raise ValueError('Invalid subtree mode "%s"' % mode)
        Pv = self.positive_pass()
        hooks = self.hooks_by_tensor()
        producer = {c.out: k for k, c in enumerate(self.calls)}
        G = {seed_tensor: seed.detach().clone().float()}
        P, names = [], []
        eps32 = eps

        def fire(t):
            g = G[t]
            for (k, j) in hooks.get(t, []):
                c = self.calls[k]
                jj = len(c.ins) - 1  # late-binding closure: last input's (a, x)   (whitebox.py:379-381)
                a = F.relu(self.T[c.ins[jj]])
                x = F.relu(Pv[c.ins[jj]])
                zh = F.relu(g)
                p = a * zh
                p_prior = None
                if priors is not None:
                    p_prior = priors.get(len(P), None)
                if p_prior is not None:
                    p = p_prior.clone()
                P.append(p)
                names.append(c.name)
                name = c.name
                if mode == 'affineonly':
                    if _is_affine(name):
                        g = p / (x + eps32)
                    # else: None -> gradient unchanged
                elif mode == 'affineonly_with_prior':
                    if p_prior is not None:
                        zh = (p_prior > 0) * g
                        p = (p_prior > 0) * p
                    if _is_affine(name):
                        g = p / (x + eps32)
                    else:
                        g = zh
                elif mode == 'norelu':
                    if (('MaxPool' in name) or ('ReLU' in name)) and p_prior is not None:
                        pass
                    else:
                        g = p / (x + eps32)
                else:  # 'all'
                    g = p / (x + eps32)
            G[t] = g

        # seed tensor may itself be hooked (it is the input of later, skipped calls only if those
        # calls are not on the tape -- the tape always ends at the seeded tensor)
        for k in range(len(self.calls) - 1, -1, -1):
            c = self.calls[k]
            if c.out not in G:
                continue
            fire(c.out)
            g_out = G.pop(c.out)
            ins = [self.T[i].detach().clone().requires_grad_(True) for i in c.ins]
            with torch.enable_grad():
                out = c.fn(ins, c.hooked)  # hooked modules: VJP with W+ at the TRUE forward point (:371-377)
                gins = torch.autograd.grad(out, ins, g_out, allow_unused=True)
            for i, gi in zip(c.ins, gins):
                if gi is None:
                    continue
                if i in G:
                    G[i] = G[i] + gi
                else:
                    G[i] = gi
        fire(0)
        return P, names


    def true_gradients(self, seed_tensor, seed):
        """The `dA` list of whitebox.py:355-358: in 'activation' mode every hooked module input carries a plain tensor hook
        `_savegrad` that records its (true-weight) gradient; one entry per hook firing, same order as Whitebox.P."""
        hooks = self.hooks_by_tensor()
        G = {seed_tensor: seed.detach().clone().float()}
        dA = []
        for k in range(len(self.calls) - 1, -1, -1):
            c = self.calls[k]
            if c.out not in G:
                continue
            for _ in hooks.get(c.out, []):
                dA.append(G[c.out])
            g_out = G.pop(c.out)
            ins = [self.T[i].detach().clone().requires_grad_(True) for i in c.ins]
            with torch.enable_grad():
                out = c.fn(ins, False)
                gins = torch.autograd.grad(out, ins, g_out, allow_unused=True)
            for i, gi in zip(c.ins, gins):
                if gi is None:
                    continue
                G[i] = G[i] + gi if i in G else gi
        for _ in hooks.get(0, []):
            dA.append(G[0])
        return dA


# ---------------------------------------------------------------------------------------------
# Backbones (forward functions restated from the reference definitions)
# ---------------------------------------------------------------------------------------------

def stresnet_forward(tape, x, layers=(3, 4, 23, 3), classifier=('hooked', None), mode='classify'):
    """resnet.py:224-265 ResNet.forward with Bottleneck.forward resnet.py:129-149.

    classifier: ('hooked', None)  -> net.fc2 from params (hooked nn.Linear, resnet.py:189)
                ('triplet', W)    -> un-hooked 2x512 Linear(bias=False) (whitebox.py:93-96)
    """
    t = tape.input(x)
    t = tape.conv(t, 'conv1', stride=2, pad=3)
    t = tape.batchnorm(t, 'bn1')
    t = tape.relu_(t)
    t = tape.maxpool(t, 3, 2, 1)
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), layers)):
        stride0 = 1 if li == 0 else 2
        for b in range(blocks):
            stride = stride0 if b == 0 else 1
            pre = 'layer%d.%d.' % (li + 1, b)
            downsample = (b == 0) and (stride != 1 or inplanes != planes * 4)   # resnet.py:202
            residual = t
            o = tape.conv(t, pre + 'conv1', stride=stride, pad=0)            # resnet.py:116 stride on conv1
            o = tape.batchnorm(o, pre + 'bn1')
            o = tape.relu_(o)
            o = tape.conv(o, pre + 'conv2', stride=1, pad=1)
            o = tape.batchnorm(o, pre + 'bn2')
            o = tape.relu_(o)
            o = tape.conv(o, pre + 'conv3', stride=1, pad=0)
            o = tape.batchnorm(o, pre + 'bn3')
            if downsample:                                                       # resnet.py:210-213
                residual = tape.avgpool(t, stride, stride)
                residual = tape.concat_channels(residual, planes * 4 // inplanes - 1)
            t = tape.relu_(tape.add(o, residual))                               # resnet.py:149
            inplanes = planes * 4
    t = tape.avgpool(t, 7, 7)
    t = tape.g_flatten(t)
    t = tape.linear(t, 'fc1')
    t = tape.g_normalize(t)
    t = tape.multiply(t, 50.0)
    if mode == 'encode':
        return t
    if classifier[0] == 'hooked':
        return tape.linear(t, 'fc2')
    return tape.g_linear(t, classifier[1])


_R50_STAGES = ((2, 3, 64, 256), (3, 4, 128, 512), (4, 6, 256, 1024), (5, 3, 512, 2048))


def resnet50_128_forward(tape, x, classifier=None, mode='classify'):
    """models/resnet50_128_pytorch/resnet50_128.py:172-348; wrapper whitebox.py:210-233.

    classifier: weight (2x128) of the wrapper-level, never-hooked `fc1` (whitebox.py:216,230).
    """
    t = tape.input(x)
    t = tape.conv(t, 'conv1_7x7_s2', stride=2, pad=3)
    t = tape.batchnorm(t, 'conv1_7x7_s2_bn')
    t = tape.relu_(t)
    t = tape.maxpool(t, 3, 2, 0, True)
    for (s, nblocks, mid, outc) in _R50_STAGES:
        for b in range(1, nblocks + 1):
            pre = 'conv%d_%d' % (s, b)
            stride = 2 if (b == 1 and s > 2) else 1
            block_in = t
            o = tape.conv(t, pre + '_1x1_reduce', stride=stride, pad=0)
            o = tape.batchnorm(o, pre + '_1x1_reduce_bn')
            o = tape.relu_(o)
            o = tape.conv(o, pre + '_3x3', stride=1, pad=1)
            o = tape.batchnorm(o, pre + '_3x3_bn')
            o = tape.relu_(o)
            o = tape.conv(o, pre + '_1x1_increase', stride=1, pad=0)
            o = tape.batchnorm(o, pre + '_1x1_increase_bn')
            if b == 1:
                sc = tape.conv(block_in, pre + '_1x1_proj', stride=stride, pad=0)
                sc = tape.batchnorm(sc, pre + '_1x1_proj_bn')
            else:
                sc = block_in
            t = tape.g_add(sc, o)                 # functional torch.add(shortcut, 1, increase_bn) :187
            t = tape.relu_(t)
    t = tape.avgpool(t, 7, 1)
    t = tape.conv(t, 'feat_extract', stride=1, pad=0)
    t = tape.g_flatten(t)
    if mode == 'encode':
        return t
    return tape.g_linear(t, classifier)


def lightcnn29v2_forward(tape, x, classifier=('hooked', None), mode='classify'):
    """lightcnn.py:249-275 network_29layers_v2.forward; mfm :58-62, group :70-73, resblock :83-89."""
    def mfm(t, pre, k, pad):
        t = tape.conv(t, pre + '.filter', stride=1, pad=pad)
        t = tape.split(t)
        return tape.g_max_halves(t)

    def group(t, pre):
        t = mfm(t, pre + '.conv_a', 1, 0)
        return mfm(t, pre + '.conv', 3, 1)

    def resblock(t, pre):
        res = t
        o = mfm(t, pre + '.conv1', 3, 1)
        o = mfm(o, pre + '.conv2', 3, 1)
        return tape.add(o, res)

    def pool(t):
        return tape.g_add(tape.maxpool(t, 2, 2), tape.avgpool(t, 2, 2))   # lightcnn.py:252

    t = tape.input(x)
    t = mfm(t, 'conv1', 5, 2)
    t = pool(t)
    for i in range(1):
        t = resblock(t, 'block1.%d' % i)
    t = group(t, 'group1')
    t = pool(t)
    for i in range(2):
        t = resblock(t, 'block2.%d' % i)
    t = group(t, 'group2')
    t = pool(t)
    for i in range(3):
        t = resblock(t, 'block3.%d' % i)
    t = group(t, 'group3')
    for i in range(4):
        t = resblock(t, 'block4.%d' % i)
    t = group(t, 'group4')
    t = pool(t)
    t = tape.g_flatten(t)
    t = tape.linear(t, 'fc')
    # F.dropout(training=False) is the identity (lightcnn.py:273)
    if mode == 'encode':
        return t
    if classifier[0] == 'hooked':
        return tape.linear(t, 'fc2')
    return tape.g_linear(t, classifier[1])


FORWARDS = {
    'stresnet101': lambda tape, x, classifier, mode: stresnet_forward(tape, x, (3, 4, 23, 3), classifier, mode),
    'stresnet_mini': lambda tape, x, classifier, mode: stresnet_forward(tape, x, (1, 1, 1, 1), classifier, mode),
    'resnet50_128': lambda tape, x, classifier, mode: resnet50_128_forward(
        tape, x, None if classifier is None else classifier[1], mode),
    'lightcnn29v2': lambda tape, x, classifier, mode: lightcnn29v2_forward(tape, x, classifier, mode),
}


# ---------------------------------------------------------------------------------------------
# Whitebox-level functions
# ---------------------------------------------------------------------------------------------

def mwp_to_saliency(P, eps=1e-16, blur_radius=2):
    """whitebox.py:448-460, ebp_ver 6 branch: skimage.filters.gaussian(img, 2) == scipy gaussian_filter
    (mode='nearest', truncate=4.0), clamp at 0, divide by max(sum, eps)."""
    img = scipy.ndimage.gaussian_filter(np.asarray(P, dtype=np.float32), blur_radius, mode='nearest', truncate=4.0)
    img = np.maximum(0, img)
    img /= (max(img.sum(), eps))
    return img


def mwp_to_saliency_uint8(P, eps=1e-16, blur_radius=2):
    """whitebox.py:451-454, ebp_ver != 6: uint8 min-max -> PIL GaussianBlur(radius) -> uint8 min-max."""
    import PIL.Image
    import PIL.ImageFilter
    img = np.asarray(P)
    img = np.uint8(255 * ((img - np.min(img)) / (eps + (np.max(img) - np.min(img)))))
    img = np.array(PIL.Image.fromarray(img).filter(PIL.ImageFilter.GaussianBlur(radius=blur_radius)))
    img = np.uint8(255 * ((img - np.min(img)) / (eps + (np.max(img) - np.min(img)))))
    return img


class OracleWhitebox(object):
    """The method surface of whitebox.Whitebox that is on the hot path, evaluated by the tape."""

    def __init__(self, arch, params, classifier=('hooked', None), ebp_subtree_mode='affineonly_with_prior',
                 eps=1e-16, with_bias=False, ebp_version=6):
        self.ebp_ver = ebp_version
        self.arch = arch
        self.params = {k: v.detach().clone().float() for k, v in params.items()}
        self.classifier = classifier
        self.mode = ebp_subtree_mode
        self.eps = eps
        self.with_bias = with_bias
        self.P = []
        self.P_layername = []

    def set_triplet_classifier(self, x_mate, x_nonmate):
        self.classifier = ('triplet', torch.cat((x_mate, x_nonmate), dim=0).detach().clone().float())

    def _run(self, x, mode):
        tape = Tape(self.params, self.with_bias)
        out = FORWARDS[self.arch](tape, x, self.classifier, mode)
        return tape, out

    def encode(self, x):
        tape, out = self._run(x, 'encode')
        return tape.T[out]

    def classify(self, x):
        tape, out = self._run(x, 'classify')
        return tape.T[out]

    def ebp(self, x, Pn, mwp=False, priors=None):
        """whitebox.py:482-504"""
        tape, out = self._run(x, 'classify')
        self.P, self.P_layername = tape.backward(out, Pn, self.mode, self.eps, priors)
        P = np.squeeze(np.sum(self.P[-2].detach().cpu().numpy(), axis=1)).astype(np.float32)
        return self._sal(P) if not mwp else P

    def _sal(self, P):
        return mwp_to_saliency(P, self.eps) if self.ebp_ver == 6 else mwp_to_saliency_uint8(P, self.eps)

    def num_classes(self):
        if self.classifier is not None and self.classifier[1] is not None:
            return int(self.classifier[1].shape[0])
        return int(self.params['fc2.weight'].shape[0])

    def _onehot(self, x, k):
        C = self.num_classes()
        assert 0 <= k < C
        P0 = torch.zeros((1, C))
        P0[0][k] = 1.0
        return P0

    def contrastive_ebp(self, x, k_pos, k_neg):
        """whitebox.py:506-527"""
        self.ebp(x, self._onehot(x, k_pos))
        P_mate = self.P
        self.ebp(x, self._onehot(x, k_neg))
        P_nonmate = self.P
        mwp_mate = P_mate[-2] / torch.sum(P_mate[-2])
        mwp_nonmate = P_nonmate[-2] / torch.sum(P_nonmate[-2])
        c = np.squeeze(np.sum(F.relu(mwp_mate - mwp_nonmate).numpy(), axis=1).astype(np.float32))
        return self._sal(c)

    def truncated_contrastive_ebp(self, x, k_pos, k_neg, percentile=20):
        """whitebox.py:529-558"""
        self.ebp(x, self._onehot(x, k_pos))
        P_mate = self.P
        self.ebp(x, self._onehot(x, k_neg))
        P_nonmate = self.P
        mwp_mate = P_mate[-2] / torch.sum(P_mate[-2])
        mwp_nonmate = P_nonmate[-2] / torch.sum(P_nonmate[-2])
        (s, idx) = torch.sort(torch.flatten(mwp_mate.clone()))
        cs = torch.cumsum(s, 0)
        mask = torch.zeros(s.shape)
        mask[idx] = (cs >= (percentile / 100.0) * cs[-1]).type(torch.FloatTensor)
        mask = mask.reshape(mwp_mate.shape)
        t = F.relu(mask * mwp_mate - mask * mwp_nonmate)
        c = np.squeeze(np.sum(t.numpy(), axis=1).astype(np.float32))
        return self._sal(c)

    def _scale_normalized(self, img):
        img = np.float32(img)
        return (img - np.min(img)) / (self.eps + (np.max(img) - np.min(img)))

    def layerwise_ebp(self, x, k_layer, mode='argmax', k_element=None, k_poschannel=0, mwp=True):
        """whitebox.py:561-581"""
        P0 = self._onehot(x, k_poschannel)
        self.ebp(x, P0)
        P_mate = self.P
        if mode == 'argmax':
            prior = P_mate[k_layer] * (1.0 - torch.ne(P_mate[k_layer], torch.max(P_mate[k_layer])).float())
        elif mode == 'elementwise':
            assert k_element is not None
            Pf = (0 * (P_mate[k_layer].detach().clone())).flatten()
            Pf[k_element] = P_mate[k_layer].flatten()[k_element]
            prior = Pf.reshape(P_mate[k_layer].shape)
        else:
            raise ValueError('invalid layerwise EBP mode "%s"' % mode)
        return self.ebp(x, 0.0 * P0, mwp=mwp, priors={int(k_layer): prior})

    def layerwise_contrastive_ebp(self, x, k_poschannel, k_negchannel, k_layer, mode='copy', percentile=80, k_element=None, gradlayer=None, mwp=False):
        """whitebox.py:584-645 (deprecated there; restated for completeness)."""
        self.ebp(x, self._onehot(x, k_poschannel))
        P_mate = self.P
        P0 = self._onehot(x, k_negchannel)
        self.ebp(x, P0)
        P_nonmate = self.P
        Pm, Pn = P_mate[k_layer], P_nonmate[k_layer]
        C = F.relu(Pm - Pn)
        argmax = lambda t: torch.mul(t, 1.0 - torch.ne(t, torch.max(t)).type(torch.FloatTensor))          # noqa: E731
        product = lambda: torch.sqrt(torch.mul(Pm.type(torch.DoubleTensor), C.type(torch.DoubleTensor))).type(torch.FloatTensor)     # noqa: E731
        if mode == 'copy':
            prior = C
        elif mode == 'mean':
            prior = 0.5 * (Pm + C)
        elif mode == 'product':
            prior = product()
        elif mode == 'argmax':
            prior = argmax(C)
        elif mode == 'argmax_product':
            prior = argmax(product())
        elif mode == 'percentile' or mode == 'percentile_argmax':
            (srt, idx) = torch.sort(torch.flatten(Pm.clone()))
            cs = torch.cumsum(srt, 0)
            mask = torch.zeros(srt.shape)
            mask[idx] = (cs >= (percentile / 100.0) * cs[-1]).type(torch.FloatTensor)
            prior = torch.mul(mask.reshape(Pm.shape), C.type(torch.FloatTensor)).clone()
            if mode == 'percentile_argmax':
                prior = argmax(prior)
        elif mode == 'elementwise':
            assert gradlayer[k_layer].shape == Pm.shape
            Pf = (0 * C.detach().clone()).flatten()
            Pf[k_element] = C.flatten()[k_element]
            prior = Pf.reshape(C.shape)
        else:
            raise ValueError('unknown contrastive ebp mode "%s"' % mode)
        k = int(k_layer) if int(k_layer) >= 0 else int(k_layer) + len(P_mate)
        return self.ebp(x, 0.0 * P0, mwp=mwp, priors={k: prior})

    def weighted_subtree_ebp(self, x, k_poschannel, k_negchannel, topk=1, do_max_subtree=False,
                             do_mated_similarity_gating=True, subtree_mode='norelu', do_mwp_to_saliency=True):
        """whitebox.py:647-737."""
        self.mode = subtree_mode
        tape, out = self._run(x, 'classify')
        C = tape.T[out].shape[1]
        y = tape.T[out]
        if not do_mated_similarity_gating:
            g = torch.softmax(y, dim=1).clone()
            g[0, 0] -= 1.0                                  # d cross_entropy(y, [0]) / dy
            gradlist_ce = tape.true_gradients(out, g)
        e0 = torch.zeros((1, C)); e0[0, 0] = 1.0
        e1 = torch.zeros((1, C)); e1[0, 1] = 1.0
        gradlist_mated = tape.true_gradients(out, e0)
        gradlist_nonmated = tape.true_gradients(out, e1)
        P_img, P_subtree, P_subtree_idx = [], [], []
        n_layers = len(gradlist_mated)
        for k in range(0, n_layers - 1):
            if do_mated_similarity_gating:
                v = torch.mul(gradlist_mated[k] >= 0, -gradlist_nonmated[k])
            else:
                v = torch.mul(gradlist_ce[k] < 0, -gradlist_nonmated[k])
            P_subtree.append(float(torch.max(v)))
            P_subtree_idx.append(torch.argmax(v))
        k_subtree = np.argsort(np.array(P_subtree))
        for k in k_subtree:
            P_img.append(self.layerwise_ebp(x, k_layer=k, k_poschannel=k_poschannel, k_element=P_subtree_idx[k], mode='elementwise'))
        k_valid = [np.max(P) > 0 for P in P_img]
        k_subtree_valid = [k for (k, v) in zip(k_subtree, k_valid) if v == True and k != 1][-topk:]   # noqa: E712
        if len(k_subtree_valid) == 0:
            raise RuntimeError('Failed to calculate valid subtrees.')
        P_img_valid = [p for (p, k, v) in zip(P_img, k_subtree, k_valid) if v == True and k != 1][-topk:]  # noqa: E712
        P_subtree_valid = [P_subtree[k] for k in k_subtree_valid]
        sn = self._scale_normalized(P_subtree_valid)
        P_subtree_valid_norm = sn if not np.sum(sn) == 0 else np.ones_like(P_subtree_valid)
        stack = np.dstack([float(w) * np.array(P) * (1.0 / (np.max(P) + 1E-12)) for (w, P) in zip(P_subtree_valid_norm, P_img_valid)])
        smap = np.max(stack, axis=2) if do_max_subtree else np.sum(stack, axis=2)
        if self.ebp_ver != 6:                                # whitebox.py:726-727 (convert_saliency_uint8)
            smap = np.uint8(255 * ((smap - np.min(smap)) / (self.eps + (np.max(smap) - np.min(smap)))))
        else:
            smap /= max(smap.sum(), self.eps)
        return (self._sal(smap) if do_mwp_to_saliency else smap,
                [self._sal(P) if do_mwp_to_saliency else P for P in P_img_valid],
                P_subtree_valid, k_subtree_valid)
