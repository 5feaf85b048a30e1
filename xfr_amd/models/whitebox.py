"""Drop-in mirror of python/xfr/models/whitebox.py for the excitation-backprop hot path.

Same classes, method names, argument meaning, return types and error behaviour as the reference
(`WhiteboxNetwork`, `WhiteboxSTResnet`, `WhiteboxLightCNN`, `Whitebox_resnet50_128`, `Whitebox`), but no
hooks and no autograd: every call is a few launches of the HIP engine (xfr_amd/csrc) through the C ABI of
include/xfr_amd.h.  Additive API: `*_batch` methods that run many independent probes in one engine call
(the reference is batch-1 only for contrastive EBP: whitebox.py:512,524).

Also provided ("next" row of the scope table): layerwise_ebp / weighted_subtree_ebp (whitebox.py:561-581, 647-737).
Not provided (outside the hot path): layerwise_contrastive_ebp (deprecated by the reference itself,
whitebox.py:587-588), Whitebox_senet50_256 (unsupported by the reference itself:
whitebox.py:402-403), DataFrame / image_loader inputs of `embeddings`.
"""
import warnings

import numpy as np
import PIL
import PIL.Image
import PIL.ImageFilter
import torch

from ..engine import Engine
from .lightcnn import lightcnn_preprocess
from .resnet import MEAN_RGB, convert_resnet101v4_image


def _is_dataframe(obj):
    """pandas.DataFrame without importing pandas unless the caller already did (whitebox.py:752)."""
    import sys
    pd = sys.modules.get('pandas')
    return pd is not None and isinstance(obj, pd.DataFrame)


def _triplet_rows(x_mate, x_nonmate):
    """The 2 x D weight of the triplet classifier, WHERE THE ENCODINGS ARE: encodings that encode() left on the device stay there -- a `.cpu()` here made
    every demo sequence (demo/test_whitebox.py:124-133: two encodes, set_triplet_classifier, contrastive_ebp) wait twice for the device before the sweep
    was even enqueued."""
    a, b = torch.as_tensor(x_mate).detach().float().reshape(1, -1), torch.as_tensor(x_nonmate).detach().float().reshape(1, -1)
    return torch.cat((a, b.to(a.device)), dim=0)


class _TripletClassifier(object):
    """Stand-in for the nn.Linear(D, 2, bias=False) that set_triplet_classifier installs
    (whitebox.py:95-96,123-124,220).  It is created after hook registration, hence un-hooked."""

    def __init__(self, weight):
        self.weight = weight.detach().clone().float()
        self.out_features = int(weight.shape[0])
        self.in_features = int(weight.shape[1])
        self.bias = None


class WhiteboxNetwork(object):
    """whitebox.py:25-85.  `net` is an xfr_amd backbone (parameter store + layer program)."""

    default_max_batch = 32

    def __init__(self, net):
        self.net = net
        self.net.eval()
        self._engine = None
        self._engine_key = None
        self._classifier = None      # _TripletClassifier or None (hooked classifier inside the program)

    # -- engine plumbing -----------------------------------------------------------------------------------
    def engine(self, min_batch=1):
        dev = self.net.device
        want = max(int(min_batch), self.default_max_batch)
        key = (str(dev), id(self.net))
        if self._engine is None or self._engine_key != key or self._engine.max_batch < min_batch:
            options = {}
            if self._engine is not None:
                if self._engine_key == key:
                    options = dict(self._engine.options)     # set_pipeline / set_tail_balance / set_epilogue_fusion survive
                    warnings.warn('xfr_amd: rebuilding the engine for batch %d (it was sized for %d): the weights are packed and '
                                  'uploaded again; size it once with default_max_batch' % (want, self._engine.max_batch))
                self._engine.close()
            self._program = self.net.build_program()
            self._engine = Engine(self._program, want, dev)
            self._engine.apply_options(options)
            spec = self.u8_preprocess_spec()
            if spec is not None:
                self._engine.set_u8_preprocess(*spec)
            self._engine_key = key
        if self._engine.loaded_version != self.net.version:
            self._engine.load_weights(self.net.state_dict())
            self._engine.loaded_version = self.net.version
        return self._engine

    def _mark(self, name):
        self.engine()
        return self._program.marks[name]

    def u8_preprocess_spec(self):
        """(kind, channels, mean, weight) of the pixel arithmetic in `preprocess` for the engine's uint8 entry points (xfr_forward_u8), or None."""
        return None

    def encode_u8(self, images_u8):
        """Additive: encode() on uint8 N x H x W x C crops (already resized / cropped like `preprocess` does it): the images travel to the device as
        uint8 and the engine applies the pixel arithmetic of `preprocess` itself, bit for bit (include/xfr_amd.h)."""
        x = torch.as_tensor(images_u8)
        eng = self.engine(x.shape[0])
        return eng.forward_u8(x, self._mark('encode')).reshape(x.shape[0], -1)

    def _cls_weight(self, device):
        return self._classifier.weight.to(device)

    def seed_for(self, Pn, n):
        """Gradient seed of `Xn.backward(Pn)` (whitebox.py:498): (tensor id, N x D seed)."""
        eng = self.engine(n)
        Pn = torch.as_tensor(Pn, dtype=torch.float32)
        if Pn.dim() == 1:
            Pn = Pn.unsqueeze(0)
        if Pn.shape[0] == 1 and n > 1:
            Pn = Pn.expand(n, -1)
        Pn = Pn.to(eng.device)
        if self._classifier is None:
            return self._program.marks['classify'], Pn.contiguous()
        return self._program.marks['encode'], (Pn @ self._cls_weight(eng.device)).contiguous()

    # -- reference surface ---------------------------------------------------------------------------------
    def _layer_visitor(self, f_forward=None, f_preforward=None, net=None):
        """whitebox.py:34-56.  There are no hooks to register here; returns the layer list for inspection."""
        from ..program import LAYER_NAMES, OpKind
        prog = self.net.build_program()
        return [{'name': LAYER_NAMES[OpKind(o.kind)], 'hooks': []} for o in prog.ops if o.kind < OpKind.G_ADD]

    def encode(self, x):
        eng = self.engine(x.shape[0])
        out = eng.forward(x, self._program.marks['encode'])
        return out.reshape(out.shape[0], -1).to(x.device)

    def classify(self, x):
        eng = self.engine(x.shape[0])
        if self._classifier is None:
            out = eng.forward(x, self._program.marks['classify'])
            return out.reshape(out.shape[0], -1).to(x.device)
        enc = eng.forward(x, self._program.marks['encode']).reshape(x.shape[0], -1)
        return (enc @ self._cls_weight(eng.device).t()).to(x.device)

    def clear(self):
        """whitebox.py:66-71: zero parameter gradients.  The engine keeps no gradient state between calls."""
        return None

    def set_triplet_classifier(self, x_mate, x_nonmate):
        raise NotImplementedError

    def num_classes(self):
        raise NotImplementedError

    def preprocess(self, im):
        raise NotImplementedError


class WhiteboxSTResnet(WhiteboxNetwork):
    """whitebox.py:87-110"""

    def set_triplet_classifier(self, x_mate, x_nonmate):
        w = _triplet_rows(x_mate, x_nonmate)
        self._classifier = _TripletClassifier(w)
        self.net.fc2 = self._classifier

    def num_classes(self):
        return self._classifier.out_features if self._classifier is not None else self.net.num_classes

    def preprocess(self, im):
        return convert_resnet101v4_image(im.resize((224, 224))).unsqueeze(0)

    def u8_preprocess_spec(self):
        return ('sub_mean', 3, tuple(float(v) for v in MEAN_RGB), None)            # resnet.py:23-37


class WhiteboxLightCNN(WhiteboxNetwork):
    """whitebox.py:113-159"""
    default_max_batch = 32

    def __init__(self, net):
        WhiteboxNetwork.__init__(self, net)
        self.f_preprocess = lightcnn_preprocess()

    def set_triplet_classifier(self, x_mate, x_nonmate):
        w = _triplet_rows(x_mate, x_nonmate)
        self._classifier = _TripletClassifier(w)
        self.net.fc2 = self._classifier

    def num_classes(self):
        return self._classifier.out_features if self._classifier is not None else self.net.num_classes

    def preprocess(self, im):
        return self.f_preprocess(im)

    def u8_preprocess_spec(self):
        return ('luminance', 3, None, (0.2125, 0.7154, 0.0721))                        # lightcnn.py:19-25 (skimage rgb2gray)


class Whitebox_resnet50_128(WhiteboxNetwork):
    """whitebox.py:210-258: the 2-way classifier `fc1` lives on the wrapper and is never hooked."""

    def __init__(self, net):
        WhiteboxNetwork.__init__(self, net)
        g = torch.Generator().manual_seed(0)
        self._classifier = _TripletClassifier((torch.rand((2, 128), generator=g) * 2 - 1) / np.sqrt(128.0))
        self.fc1 = self._classifier

    def set_triplet_classifier(self, x_mate, x_nonmate):
        w = _triplet_rows(x_mate, x_nonmate)
        self._classifier = _TripletClassifier(w)
        self.fc1 = self._classifier

    def num_classes(self):
        return 128 if self.fc1 is None else self.fc1.out_features

    def preprocess(self, img):
        """whitebox.py:235-258"""
        mean = (131.0912, 103.8827, 91.4953)
        short_size = 224.0
        crop_size = (224, 224, 3)
        im_shape = np.array(img.size)
        img = img.convert('RGB')
        ratio = float(short_size) / np.min(im_shape)
        img = img.resize(size=(int(np.ceil(im_shape[0] * ratio)), int(np.ceil(im_shape[1] * ratio))),
                         resample=PIL.Image.BILINEAR)
        x = np.array(img)
        newshape = x.shape[:2]
        h_start = (newshape[0] - crop_size[0]) // 2
        w_start = (newshape[1] - crop_size[1]) // 2
        x = x[h_start:h_start + crop_size[0], w_start:w_start + crop_size[1]]
        x = x - mean
        return torch.from_numpy(x.transpose(2, 0, 1).astype(np.float32)).unsqueeze(0)

    def u8_preprocess_spec(self):
        return ('sub_mean', 3, (131.0912, 103.8827, 91.4953), None)                    # whitebox.py:238,256


class _PList(object):
    """`Whitebox.P` (whitebox.py:294,394): the MWP tensors of the last sweep in firing order.  The reference clones every one
    of them on every sweep; the engine hands back the channel-pooled P[-2] its callers read (whitebox.py:499,524) and recomputes
    any entry on demand with one more sweep that stores that firing (xfr_ebp_store_firing).  len(P) counts the image hook
    P[-1] like the reference does; its tensor is the first layer's backward-data pass (a gather kernel run on demand, round 4)."""

    def __init__(self, wb, x, seed_tensor, seed, n_firings):
        self._wb, self._x, self._seed_tensor, self._seed = wb, x, seed_tensor, seed
        self._cache = {}
        self._n = n_firings + 1

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        k = int(i)
        if k < 0:
            k += self._n
        if not (0 <= k < self._n):
            raise IndexError('P index %d out of range (%d entries)' % (i, self._n))
        if k not in self._cache:
            eng = self._wb._engine(self._x.shape[0])
            P = eng.ebp_firing(self._x, self._seed_tensor, self._seed, k)
            # the reference flattens before its Linear layers (resnet.py:245, lightcnn.py:258): those entries are N x D
            if self._wb.P_layername[k].startswith('Linear(') or (P.shape[2] == 1 and P.shape[3] == 1):
                P = P.reshape(P.shape[0], -1)
            self._cache[k] = P
        return self._cache[k]


class Whitebox(object):
    """whitebox.py:261-304."""

    def __init__(self, net, ebp_version=None, with_bias=None, eps=1E-16, ebp_subtree_mode='affineonly_with_prior'):
        assert (isinstance(net, WhiteboxNetwork))
        self.net = net
        self.eps = eps
        self.layerlist = None
        self.ebp_ver = ebp_version
        if self.ebp_ver is None:
            self.ebp_ver = 6
        elif self.ebp_ver < 4:
            raise RuntimeError('ebp version, if set, must be at least 4')
        self.convert_saliency_uint8 = (self.ebp_ver != 6)
        if with_bias is not None:
            self._ebp_with_bias = with_bias
        else:
            self._ebp_with_bias = self.ebp_ver == 11
        self.dA = []
        self.A = []
        self.X = []
        self.P = []
        self.P_prior = []
        self.P_layername = []
        self.batch_size = 32
        self._ebp_mode = 'disable'
        self.layerlist = self.net._layer_visitor()
        if ebp_subtree_mode not in ('affineonly', 'affineonly_with_prior', 'norelu', 'all'):
            raise ValueError('Invalid subtree mode "%s"' % ebp_subtree_mode)
        self._ebp_subtree_mode = ebp_subtree_mode
        self.debug_trace = False

    # nn.Module-ish no-ops used by the reference's callers (eval/create_wbnet.py:38-42)
    def to(self, device):
        self.net.net.to(device)
        return self

    def eval(self):
        return self

    def _engine(self, n=1):
        eng = self.net.engine(n)
        eng.set_mode(self._ebp_subtree_mode, self.eps, self._ebp_with_bias)
        return eng

    # -- helpers ---------------------------------------------------------------------------------------------
    def _float32_to_uint8(self, img):
        return np.uint8(255 * ((img - np.min(img)) / (self.eps + (np.max(img) - np.min(img)))))

    def _scale_normalized(self, img):
        img = np.float32(img)
        return (img - np.min(img)) / (self.eps + (np.max(img) - np.min(img)))

    def _mwp_to_saliency(self, P, blur_radius=2):
        """whitebox.py:448-460.  ebp_ver 6: HIP blur (gaussian sigma=2, nearest) + clamp + /sum.  Other versions:
        the uint8 PIL path of the reference, on the host (not on the hot path)."""
        img = P
        if self.convert_saliency_uint8:
            img = np.uint8(255 * ((img - np.min(img)) / (self.eps + (np.max(img) - np.min(img)))))
            img = np.array(PIL.Image.fromarray(img).filter(PIL.ImageFilter.GaussianBlur(radius=blur_radius)))
            img = np.uint8(255 * ((img - np.min(img)) / (self.eps + (np.max(img) - np.min(img)))))
            return img
        if blur_radius != 2:
            raise ValueError('xfr_amd implements the reference blur radius (2) only')
        eng = self._engine()
        t = torch.as_tensor(np.asarray(P, dtype=np.float32))
        squeeze = (t.dim() == 2)
        if squeeze:
            t = t.unsqueeze(0)
        cap = max(1, 2 * eng.max_batch)            # the engine's blur scratch holds 2 x max_batch maps
        out = torch.cat([eng.mwp_to_saliency(t[i:i + cap]) for i in range(0, t.shape[0], cap)], dim=0).cpu().numpy()
        return out[0] if squeeze else out

    def _layernames(self, seed_tensor):
        key = (self._ebp_subtree_mode, int(seed_tensor))
        cache = self.__dict__.setdefault('_layername_cache', {})
        if key not in cache:
            cache[key] = self.net._program.layernames(self._ebp_subtree_mode, seed_tensor)
        return list(cache[key])

    def _clear(self):
        (self.P, self.P_layername, self.dA, self.A, self.X) = ([], [], [], [], [])
        self.net.clear()

    # -- EBP -------------------------------------------------------------------------------------------------
    def ebp(self, x, Pn, mwp=False):
        """Excitation backprop (whitebox.py:482-504).  x: N x C x U x V; Pn: N x num_classes (or 1 x ...)."""
        n = x.shape[0]
        eng = self._engine(n)
        self._clear()
        seed_tensor, seed = self.net.seed_for(Pn, n)
        if self.debug_trace:
            eng.set_trace(True)
        _, pooled = eng.ebp(x, seed_tensor, seed.unsqueeze(0), want_mwp=False, want_pooled=True)
        # whitebox.py:393: str(module) of the hooked module behind every entry of P, the image hook (P[-1]) last -- built from the layer program
        # (Program.layernames), identical to the reference's list under the same torch (tests/test_program.py, golden_layernames.npz)
        self.P_layername = self._layernames(seed_tensor)
        if self.debug_trace:
            sums, names, nf = eng.get_trace()
            assert names == [s.split('(')[0] for s in self.P_layername[:nf]]
            self.P_trace = sums[:nf * n].reshape(nf, n).copy()
            eng.set_trace(False)
        self.P = _PList(self, x, seed_tensor, seed.unsqueeze(0), len(self.P_layername) - 1)
        P = np.squeeze(pooled[0].cpu().numpy()).astype(np.float32)
        return self._mwp_to_saliency(P) if not mwp else P

    def _class_seeds(self, n, k_pos, k_neg):
        assert (k_pos >= 0 and k_pos < self.net.num_classes())
        assert (k_neg >= 0 and k_neg < self.net.num_classes())
        C = self.net.num_classes()
        P0 = torch.zeros((1, C))
        P0[0][k_pos] = 1.0
        P1 = torch.zeros((1, C))
        P1[0][k_neg] = 1.0
        t0, s0 = self.net.seed_for(P0, n)
        t1, s1 = self.net.seed_for(P1, n)
        return t0, torch.stack((s0, s1), dim=0)

    def _squeeze(self, sal):
        out = sal.cpu().numpy().astype(np.float32)
        return out[0] if out.shape[0] == 1 else out

    def contrastive_ebp(self, img_probe, k_poschannel, k_negchannel):
        """Contrastive excitation backprop (whitebox.py:506-527); per sample when N > 1."""
        n = img_probe.shape[0]
        eng = self._engine(n)
        seed_tensor, seeds = self._class_seeds(n, k_poschannel, k_negchannel)
        if self.convert_saliency_uint8:      # ebp_version != 6: uint8 + PIL blur on the host (whitebox.py:451-454)
            raw = eng.contrastive(img_probe, seed_tensor, seeds, None, raw=True).cpu().numpy()
            out = np.stack([self._mwp_to_saliency(r) for r in raw])
            return out[0] if n == 1 else out
        return self._squeeze(eng.contrastive(img_probe, seed_tensor, seeds, None))

    def truncated_contrastive_ebp(self, img_probe, k_poschannel, k_negchannel, percentile=20):
        """Truncated contrastive excitation backprop (whitebox.py:529-558)."""
        n = img_probe.shape[0]
        eng = self._engine(n)
        seed_tensor, seeds = self._class_seeds(n, k_poschannel, k_negchannel)
        if self.convert_saliency_uint8:
            raw = eng.contrastive(img_probe, seed_tensor, seeds, float(percentile), raw=True).cpu().numpy()
            out = np.stack([self._mwp_to_saliency(r) for r in raw])
            return out[0] if n == 1 else out
        return self._squeeze(eng.contrastive(img_probe, seed_tensor, seeds, float(percentile)))

    # -- additive batched API --------------------------------------------------------------------------------
    def contrastive_triplet_ebp_batch(self, img_probes, x_mates, x_nonmates, percentile=None):
        """N independent (mate, non-mate, probe) triplets in one call: for sample i equivalent to
        set_triplet_classifier(x_mates[i], x_nonmates[i]); (truncated_)contrastive_ebp(img_probes[i:i+1], 0, 1).
        x_mates / x_nonmates: N x D classifier rows (already scaled, e.g. encoding/2500:
        demo/test_whitebox.py:129).  Returns N x H x W float32 (torch, on the engine device)."""
        n = img_probes.shape[0]
        eng = self._engine(n)
        prog_marks = self.net._program.marks
        seeds = torch.stack((torch.as_tensor(x_mates, dtype=torch.float32).reshape(n, -1),
                             torch.as_tensor(x_nonmates, dtype=torch.float32).reshape(n, -1)), dim=0)
        return eng.contrastive(img_probes, prog_marks['encode'], seeds, percentile)

    def triplet_images_ebp_batch(self, img_probes, img_mates, img_nonmates, scale=1.0 / 2500.0, percentile=None,
                                 gallery=None, inputs_ready=False):
        """The whole triplet step of demo/test_whitebox.py:124-133 for N triplets given as images: encode the mates and
        non-mates, install scale*encoding as the per-triplet 2-way classifier, (truncated) contrastive EBP of the probes.
        Returns N x H x W float32 (torch, on the engine device)."""
        n = img_probes.shape[0]
        self.net.default_max_batch = max(self.net.default_max_batch, 2 * n)
        eng = self._engine(2 * n)
        if gallery is None:       # [mates; non-mates] as one 2N-image batch (pass `gallery` to avoid the copy)
            gallery = torch.cat((img_mates.to(eng.device), img_nonmates.to(eng.device)), dim=0)
        return eng.triplet_contrastive(img_probes, gallery, self.net._program.marks['encode'], scale, percentile, inputs_ready)

    def triplet_images_ebp_batch_u8(self, probes_u8, mates_u8, nonmates_u8, scale=1.0 / 2500.0, percentile=None, inputs_ready=False):
        """triplet_images_ebp_batch on uint8 N x H x W x C crops: a quarter of the input bytes cross the bus, the engine preprocesses."""
        n = probes_u8.shape[0]
        self.net.default_max_batch = max(self.net.default_max_batch, 2 * n)
        eng = self._engine(2 * n)
        gallery = torch.cat((torch.as_tensor(mates_u8).to(eng.device), torch.as_tensor(nonmates_u8).to(eng.device)), dim=0)
        return eng.triplet_contrastive_u8(torch.as_tensor(probes_u8), gallery, self.net._program.marks['encode'], scale, percentile, inputs_ready)

    def layerwise_ebp(self, img_probe, k_layer, mode='argmax', k_element=None, k_poschannel=0, mwp=True):
        """Layerwise excitation backprop (whitebox.py:561-581): a standard EBP sweep picks the starting node of layer
        `k_layer` (index into Whitebox.P), a second sweep with a zero seed propagates only that prior."""
        assert (k_poschannel >= 0 and k_poschannel < self.net.num_classes())
        assert img_probe.shape[0] == 1
        eng = self._engine(1)
        P0 = torch.zeros((1, self.net.num_classes()))
        P0[0][k_poschannel] = 1.0
        seed_tensor, seed = self.net.seed_for(P0, 1)
        nf = eng.firing_count(seed_tensor)
        k_layer = int(k_layer)
        if k_layer < 0:
            k_layer += nf + 1                                      # Python indexing into P, which has nf + 1 entries (:570)
        if not (0 <= k_layer <= nf):
            raise IndexError('list index out of range')
        if k_layer == nf:
            # the image hook fires after P[-2] has been recorded: with a zero seed nothing reaches P[-2] (:581 -> :499)
            c1, h1, w1 = eng.tensor_shape(1)
            P = np.zeros((h1, w1), dtype=np.float32)
            return self._mwp_to_saliency(P) if not mwp else P
        if mode == 'argmax':
            Pk = eng.ebp_firing(img_probe, seed_tensor, seed.unsqueeze(0), int(k_layer))
            prior = Pk * (1.0 - torch.ne(Pk, torch.max(Pk)).float())          # whitebox.py:572
            pooled = eng.layerwise(img_probe, seed_tensor, [int(k_layer)], dense_prior=prior[0])
        elif mode == 'elementwise':
            assert (k_element is not None)
            elems = [-1] * nf
            elems[int(k_layer)] = int(k_element)
            val = eng.ebp_capture(img_probe, seed_tensor, seed.unsqueeze(0), elems)[int(k_layer)]   # whitebox.py:575-577
            pooled = eng.layerwise(img_probe, seed_tensor, [int(k_layer)], [int(k_element)], [float(val)])
        else:
            raise ValueError('invalid layerwise EBP mode "%s"' % mode)
        P = np.squeeze(pooled[0].cpu().numpy()).astype(np.float32)
        return self._mwp_to_saliency(P) if not mwp else P

    @staticmethod
    def _contrastive_prior(P_mate, P_nonmate, mode, percentile=80, k_element=None):
        """The prior tensor layerwise_contrastive_ebp installs at its layer (whitebox.py:605-643), from the two MWP tensors of that layer."""
        import torch.nn.functional as F
        Pm, Pn = P_mate.detach().cpu().float(), P_nonmate.detach().cpu().float()
        C = F.relu(Pm - Pn)
        argmax = lambda t: torch.mul(t, 1.0 - torch.ne(t, torch.max(t)).type(torch.FloatTensor))          # noqa: E731
        product = lambda: torch.sqrt(torch.mul(Pm.type(torch.DoubleTensor), C.type(torch.DoubleTensor))).type(torch.FloatTensor)     # noqa: E731
        if mode == 'copy':
            return C
        if mode == 'mean':
            return 0.5 * (Pm + C)
        if mode == 'product':
            return product()
        if mode == 'argmax':
            return argmax(C)
        if mode == 'argmax_product':
            return argmax(product())
        if mode in ('percentile', 'percentile_argmax'):
            assert (percentile >= 0 and percentile <= 100)
            (Pn_sorted, idx) = torch.sort(torch.flatten(Pm.clone()))
            cs = torch.cumsum(Pn_sorted, 0)
            mask = torch.zeros(Pn_sorted.shape)
            mask[idx] = (cs >= (percentile / 100.0) * cs[-1]).type(torch.FloatTensor)
            prior = torch.mul(mask.reshape(Pm.shape), C.type(torch.FloatTensor)).clone()
            return argmax(prior) if mode == 'percentile_argmax' else prior
        if mode == 'elementwise':
            P = (0 * C.detach().clone()).flatten()
            P[k_element] = C.flatten()[k_element]
            return P.reshape(C.shape)
        raise ValueError('unknown contrastive ebp mode "%s"' % mode)

    def layerwise_contrastive_ebp(self, img_probe, k_poschannel, k_negchannel, k_layer, mode='copy', percentile=80, k_element=None, gradlayer=None,
                                  mwp=False):
        """Layerwise contrastive excitation backprop (whitebox.py:584-645; deprecated by the reference in favour of weighted_subtree_ebp, kept for
        drop-in completeness): the MWP tensors of layer `k_layer` under the two one-hot seeds (two sweeps that store that firing), a prior built from
        them on the host (`_contrastive_prior`: the reference's expressions), and a zero-seeded sweep that propagates only the prior."""
        warnings.warn("layerwise_contrastive_ebp is deprecated, use weighted_subtree_ebp instead")
        assert (k_poschannel >= 0 and k_poschannel < self.net.num_classes())
        assert (k_negchannel >= 0 and k_negchannel < self.net.num_classes())
        assert img_probe.shape[0] == 1
        eng = self._engine(1)
        seed_tensor, seeds = self._class_seeds(1, k_poschannel, k_negchannel)
        nf = eng.firing_count(seed_tensor)
        k = int(k_layer)
        if k < 0:
            k += nf + 1
        if not (0 <= k <= nf):
            raise IndexError('list index out of range')
        Pm = eng.ebp_firing(img_probe, seed_tensor, seeds[0:1], k)
        Pn = eng.ebp_firing(img_probe, seed_tensor, seeds[1:2], k)
        if k < nf and (self._layernames(seed_tensor)[k].startswith('Linear(') or (Pm.shape[2] == 1 and Pm.shape[3] == 1)):
            Pm, Pn = Pm.reshape(1, -1), Pn.reshape(1, -1)            # the reference's P of a flattened layer is N x D (see _PList)
        if mode == 'elementwise':
            assert (tuple(gradlayer[k_layer].shape) == tuple(Pm.shape))
        prior = self._contrastive_prior(Pm, Pn, mode, percentile, k_element)
        if k == nf:      # a prior at the image hook fires after P[-2] has been recorded: nothing reaches it (see layerwise_ebp)
            c1, h1, w1 = eng.tensor_shape(1)
            P = np.zeros((h1, w1), dtype=np.float32)
        else:
            pooled = eng.layerwise(img_probe, seed_tensor, [k], dense_prior=prior[0])
            P = np.squeeze(pooled[0].cpu().numpy()).astype(np.float32)
        return self._mwp_to_saliency(P) if not mwp else P

    def weighted_subtree_ebp(self, img_probe, k_poschannel, k_negchannel, topk=1, verbose=True, do_max_subtree=False,
                             do_mated_similarity_gating=True, subtree_mode='norelu', do_mwp_to_saliency=True, sweep_batch=None):
        """Weighted subtree EBP (whitebox.py:647-737).  Same result as the reference, computed with: one true-weight
        gradient pass for the layer weights (:652-697), ONE standard EBP sweep for all prior values (the reference repeats
        it for every layer, :567), and layerwise sweeps batched on the GPU and evaluated lazily from the heaviest layer
        downwards until `topk` valid subtrees exist (the reference sweeps all ~377 layers and keeps the last topk, :700-716)."""
        assert img_probe.shape[0] == 1
        self._ebp_subtree_mode = subtree_mode                                   # whitebox.py:651
        eng = self._engine(1)
        C = self.net.num_classes()
        onehot = lambda k: torch.nn.functional.one_hot(torch.tensor([k]), C).float()        # noqa: E731
        seed_tensor, s1 = self.net.seed_for(onehot(1), 1)                       # y[0][1].backward  (:676)
        if do_mated_similarity_gating:
            _, s0 = self.net.seed_for(onehot(0), 1)                             # y[0][0].backward  (:668)
        else:
            y = self.net.classify(img_probe).detach().cpu().float()
            g = torch.softmax(y, dim=1)
            g[0, 0] -= 1.0                                                      # d cross_entropy(y,[0])/dy  (:657,:664)
            _, s0 = self.net.seed_for(g, 1)
        _, sk = self.net.seed_for(onehot(k_poschannel), 1)
        return self._weighted_subtree(eng, img_probe, seed_tensor, s0, s1, sk, topk, verbose, do_max_subtree,
                                      do_mated_similarity_gating, do_mwp_to_saliency, sweep_batch)[0]

    def weighted_subtree_ebp_batch(self, img_probes, x_mates, x_nonmates, k_poschannel=0, topk=1, do_max_subtree=False,
                                   do_mated_similarity_gating=True, subtree_mode='norelu', do_mwp_to_saliency=True,
                                   sweep_batch=None):
        """Additive: weighted_subtree_ebp for N independent probes in shared launches.  For probe i equivalent to
        set_triplet_classifier(x_mates[i], x_nonmates[i]); weighted_subtree_ebp(img_probes[i:i+1], k_poschannel, 1 - k_poschannel...)
        with channel 0 the mate and channel 1 the non-mate.  The N forwards run once; the layer-weight pass carries 2N gradient
        streams, the capture pass N, and every round of layerwise sweeps J x N (J * N <= 2 * max_batch).  Returns a list of N
        (smap, P_img_valid, P_subtree_valid, k_subtree_valid) tuples."""
        n = img_probes.shape[0]
        self._ebp_subtree_mode = subtree_mode
        eng = self._engine(n)
        seed_tensor = self.net._program.marks['encode']
        xm = torch.as_tensor(x_mates, dtype=torch.float32).reshape(n, -1).to(eng.device)
        xn = torch.as_tensor(x_nonmates, dtype=torch.float32).reshape(n, -1).to(eng.device)
        if do_mated_similarity_gating:
            s0 = xm
        else:
            enc = self.net.encode(img_probes).to(eng.device)
            y = torch.stack(((enc * xm).sum(dim=1), (enc * xn).sum(dim=1)), dim=1)
            g = torch.softmax(y, dim=1)
            g[:, 0] -= 1.0
            s0 = g[:, 0:1] * xm + g[:, 1:2] * xn                               # g @ W_cls per probe
        sk = xm if k_poschannel == 0 else xn
        return self._weighted_subtree(eng, img_probes, seed_tensor, s0, xn, sk, topk, False, do_max_subtree,
                                      do_mated_similarity_gating, do_mwp_to_saliency, sweep_batch)

    def _weighted_subtree(self, eng, x, seed_tensor, s0, s1, sk, topk, verbose, do_max_subtree, do_mated_similarity_gating,
                          do_mwp_to_saliency, sweep_batch):
        """s0 / s1 / sk: N x D gradient seeds of the gate output, the non-mate output and the EBP channel at `seed_tensor`."""
        x, _ = eng._prep(x)                                                     # one device tensor for all phases ...
        n = x.shape[0]
        eng.hold_forward(True)                                                  # ... which share its forward pass
        try:
            w, idx = eng.subtree_weights(x, seed_tensor, torch.stack((s0, s1), dim=0), gate_ge0=do_mated_similarity_gating)
            nf = w.shape[0]
            vals = np.asarray(eng.ebp_capture(x, seed_tensor, sk.unsqueeze(0), idx)).reshape(nf, n)
            order = [np.argsort(w[:, b].astype(np.float64)) for b in range(n)]  # ascending (:697); float64 like the reference's list of Python floats (ties)
            J = int(sweep_batch or max(1, min((2 * eng.max_batch) // n, max(8, 2 * topk))))
            pos = [nf] * n
            valid = [[] for _ in range(n)]                                      # per probe: (k, P on the device), heaviest first
            rounds = 0

            def take(b, limit):
                """The next (up to `limit`) layers of probe b, heaviest first, that can yield a valid subtree.  A layer whose chosen
                element has P == 0 gives an all-zero prior, hence an all-zero map (np.max(P) > 0 fails, :706): it is invalid
                without being swept.  k == 1 is excluded by the reference (:707)."""
                ks = []
                while pos[b] > 0 and len(ks) < limit:
                    pos[b] -= 1
                    k = int(order[b][pos[b]])
                    if verbose:
                        print('[weighted_subtree_ebp][%d]: grad=%f' % (k, w[k, b]))
                    if vals[k, b] != 0 and k != 1:
                        ks.append(k)
                return ks

            while any(pos[b] > 0 and len(valid[b]) < topk for b in range(n)):
                # the first round sweeps J candidates per probe; a later round only what the probes still lack (plus a margin for sweeps that come
                # back all-zero) -- the order of evaluation, hence the selection, is the same, the last round is a fraction of the first
                lim = [J if rounds == 0 else min(J, 2 * (topk - len(valid[b])) + 2) for b in range(n)]
                rounds += 1
                Jr = max(1, max(lim[b] for b in range(n) if len(valid[b]) < topk))
                F = -np.ones((Jr, n), dtype=np.int32)
                E = np.zeros((Jr, n), dtype=np.int32)
                V = np.zeros((Jr, n), dtype=np.float32)
                todo = []
                for b in range(n):
                    ks = take(b, lim[b]) if len(valid[b]) < topk else []
                    row = {k: j for j, k in enumerate(sorted(ks))}             # ascending firing: a sweep joins at its own firing
                    for k, j in row.items():
                        F[j, b], E[j, b], V[j, b] = k, idx[k, b], vals[k, b]
                    todo.append((ks, row))
                if not any(ks for ks, _ in todo):
                    continue
                maps = eng.layerwise(x, seed_tensor, F, E, V)                    # J x n x H1 x W1, stays on the device
                alive = (maps.amax(dim=(2, 3)) > 0).cpu().numpy()                # np.max(P) > 0 per sweep (:706)
                for b, (ks, row) in enumerate(todo):
                    for k in ks:
                        if alive[row[k], b] and len(valid[b]) < topk:
                            valid[b].append((k, maps[row[k], b]))
            # ONE device-to-host copy for every valid map of every probe (was: one synchronising copy per map)
            flat = [P for vb in valid for _, P in vb]
            batched = do_mwp_to_saliency and not self.convert_saliency_uint8
            dev_maps = torch.stack(flat) if flat else None
            if dev_maps is None:
                host = np.zeros((0,) + tuple(eng.tensor_shape(1)[1:]), np.float32)
            elif batched:
                host = self._to_host_scratch(dev_maps)                          # only read by the merges below: a view of the pinned scratch
            else:
                host = dev_maps.cpu().numpy().astype(np.float32)                # returned to the caller: its own memory
            o, vh = 0, []
            for vb in valid:
                vh.append([(k, host[o + i]) for i, (k, _) in enumerate(vb)])
                o += len(vb)
            valid = vh
        finally:
            eng.hold_forward(False)
        res = [self._merge_subtrees(valid[b][::-1], [float(v) for v in w[:, b]], do_max_subtree, do_mwp_to_saliency and not batched) for b in range(n)]
        if batched and n > 0:
            # ... and ONE saliency conversion (blur, clamp, normalise: whitebox.py:456-459) for the merged maps and one for the top-k maps of every
            # probe instead of topk + 1 device round trips per probe (the conversion is per map, so batching it changes nothing).  The top-k maps are
            # converted from the copies that never left the device; only the merged maps travel host -> device.
            merged = self._mwp_to_saliency(np.stack([r[0] for r in res]))
            if dev_maps is not None:
                cap = max(1, 2 * eng.max_batch)
                top = torch.cat([eng.mwp_to_saliency(dev_maps[i:i + cap]) for i in range(0, dev_maps.shape[0], cap)], dim=0).cpu().numpy()
            o, out = 0, []
            for b, r in enumerate(res):
                k = len(r[1])                                                   # r[1] is valid[b] reversed (ascending weight)
                out.append((merged[b], [top[o + k - 1 - i] for i in range(k)], r[2], r[3]))
                o += k
            res = out
        return res

    def _to_host_scratch(self, t):
        """Device -> host through a cached pinned buffer; the result is a VIEW of that buffer (valid until the next call)."""
        n = t.numel()
        buf = self.__dict__.get('_pinned_scratch')
        if buf is None or buf.numel() < n:
            buf = torch.empty(n, dtype=torch.float32, pin_memory=True)
            self.__dict__['_pinned_scratch'] = buf
        v = buf[:n].view(t.shape)
        v.copy_(t.to(torch.float32))
        return v.numpy()

    def _merge_subtrees(self, valid, P_subtree, do_max_subtree, do_mwp_to_saliency):
        """whitebox.py:706-737 on the valid subtrees (ascending weight, like the reference's [-topk:])."""
        if len(valid) == 0:
            raise RuntimeError(
                'Failed to calculate valid subtrees. The ebp subtree mode '
                '(%s) may not support by this type of network. You may want '
                'to try the "affineonly_with_prior" ebp subtree mode.' % self._ebp_subtree_mode)
        k_subtree_valid = [k for k, _ in valid]
        P_img_valid = [P for _, P in valid]
        P_subtree_valid = [P_subtree[k] for k in k_subtree_valid]
        sn = self._scale_normalized(P_subtree_valid)
        P_subtree_valid_norm = sn if not np.sum(sn) == 0 else np.ones_like(P_subtree_valid)
        stack = np.dstack([float(wn) * np.array(P) * (1.0 / (np.max(P) + 1E-12)) for (wn, P) in zip(P_subtree_valid_norm, P_img_valid)])
        smap = np.max(stack, axis=2) if do_max_subtree else np.sum(stack, axis=2)
        if self.convert_saliency_uint8:
            smap = self._float32_to_uint8(smap)
        else:
            smap /= max(smap.sum(), self.eps)
        return (self._mwp_to_saliency(smap) if do_mwp_to_saliency else smap,
                [self._mwp_to_saliency(P) if do_mwp_to_saliency else P for P in P_img_valid],
                P_subtree_valid, k_subtree_valid)

    def ebp_subtree_mode(self):
        return self._ebp_subtree_mode

    def encode(self, x):
        return self.net.encode(x)

    def embeddings(self, images, norm=True):
        """whitebox.py:747-785: tensors / numpy arrays already in network format, or -- through xfr_amd.image_loader, the restatement of
        xfr.utils.image_loader -- an inpainting-game DataFrame or a list of file names / H x W x 3 arrays."""
        from ..image_loader import image_loader
        if _is_dataframe(images):
            imagesT = [self.convert_from_numpy(im)[0] for im in image_loader(images)]
        elif isinstance(images[0], torch.Tensor):
            assert images[0].ndim == 3
            imagesT = images
        elif isinstance(images[0], np.ndarray):
            assert images[0].shape[0] in (1, 3)
            imagesT = [torch.from_numpy(im).float() for im in images]
        else:
            imagesT = [self.convert_from_numpy(im)[0] for im in image_loader(images)]
        if not isinstance(imagesT, torch.Tensor):
            imagesT = torch.stack(list(imagesT))
        batches = torch.split(imagesT, self.batch_size, dim=0)
        # the reference moves every batch to the host as it goes (:776); here the encodings stay on the device until the last batch
        # is enqueued -- one synchronisation per call instead of one per batch (6500 masked probes per image at RISE scale)
        embeds = [self.encode(batch).detach() for batch in batches]
        embeds = torch.cat(embeds, dim=0).cpu().numpy()
        if norm:
            embeds = (embeds.reshape((embeds.shape[0], -1)) /
                      np.linalg.norm(embeds.reshape((embeds.shape[0], -1)), axis=1, keepdims=True)).reshape(embeds.shape)
        return embeds

    def convert_from_numpy(self, img):
        """whitebox.py:787-806: float RGB image (H x W x 3, range 0..1 or 0..255) or uint8 image -> network input tensor.
        Where the reference drops into pdb for out-of-range data (:797-800) this raises ValueError."""
        from ..saliency_io import resize_linear
        if isinstance(img, torch.Tensor):        # additive: a tensor is taken to be in network format already (1|N x C x U x V)
            return img if img.dim() == 4 else img.unsqueeze(0)
        if img.dtype == np.uint8:
            img = img.astype(np.float32) / 255
        if img.max() > 1 + 1e-6 and img.min() > 0 - 1e-6:
            img = img / 255
        if img.max() > 1 + 1e-6 or img.min() < 0 - 1e-6:
            raise ValueError('convert_from_numpy: image values outside [0, 1] / [0, 255]')
        img = resize_linear(img, (224, 224))
        img = (img * 255).astype(np.uint8)
        return self.net.preprocess(PIL.Image.fromarray(img).convert('RGB'))

    def preprocess_loader(self, images, returnImageIndex=False, repeats=1):
        """whitebox.py:808-825: iterate (displayable image, tensor, fn) over `images`.  The reference pulls the images through
        xfr.utils.image_loader (here: xfr_amd.image_loader): H x W x 3 arrays (fn is None for those), file names, DataFrames."""
        from ..image_loader import image_loader
        if returnImageIndex or repeats != 1:
            raise NotImplementedError('preprocess_loader: the reference itself only unpacks (image, fn) pairs (whitebox.py:817)')
        for im, fn in image_loader(images, returnFileName=True):
            yield im, self.convert_from_numpy(im)[0], fn
