"""Python handle on one HIP engine (one backbone, one device).  PyTorch tensors are only the container for
device memory and the source of the HIP stream; all arithmetic happens inside libxfr_amd.so."""
import ctypes

import numpy as np
import torch

from . import _lib
from .program import LAYER_NAMES, OpKind

MODES = {'affineonly': 0, 'affineonly_with_prior': 1, 'norelu': 2, 'all': 3}


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Engine(object):
    def __init__(self, program, max_batch, device):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError('xfr_amd: no HIP device visible; the engine has no CPU fallback')
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('xfr_amd: the engine runs on a HIP device, got %s' % (device,))
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.program = program
        self.max_batch = int(max_batch)
        self._h = ctypes.c_void_p()
        ops = program.op_array()
        c, h, w = program.in_shape
        _lib.check(self.lib.xfr_engine_create(ops, len(program.ops), len(program.weight_names), c, h, w,
                                              self.max_batch, self.device.index, ctypes.byref(self._h)))
        self._mode = None
        self.loaded_version = None
        self._pipeline = 0
        self.options = {}          # engine-level switches set through this handle (re-applied when a wrapper rebuilds the engine)

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self.lib.xfr_engine_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------
    def load_weights(self, state_dict):
        names = self.program.weight_names
        views = (_lib.TensorView * len(names))()
        keep = []
        for i, n in enumerate(names):
            if n not in state_dict:
                raise KeyError('xfr_amd: parameter "%s" missing from the state_dict' % n)
            t = state_dict[n].detach().to('cpu', torch.float32).contiguous()
            keep.append(t)
            views[i].data = t.data_ptr()
            views[i].numel = t.numel()
        _lib.check(self.lib.xfr_engine_load_weights(self._h, views, len(names)))

    def weight_arena(self):
        """The packed parameter arena as a uint8 CUDA tensor view (for torch.distributed.broadcast)."""
        p = ctypes.c_void_p()
        nbytes = ctypes.c_size_t()
        _lib.check(self.lib.xfr_engine_weight_arena(self._h, ctypes.byref(p), ctypes.byref(nbytes)))

        class _Arr(object):
            pass
        a = _Arr()
        a.__cuda_array_interface__ = {'shape': (nbytes.value,), 'typestr': '|u1', 'data': (p.value, False),
                                      'version': 2}
        with torch.cuda.device(self.device):
            t = torch.as_tensor(a, device=self.device)
        return t

    def mark_weights_loaded(self):
        _lib.check(self.lib.xfr_engine_mark_weights_loaded(self._h))

    def set_mode(self, subtree_mode, eps=1e-16, with_bias=False):
        if subtree_mode not in MODES:
            raise ValueError('Invalid subtree mode "%s"' % subtree_mode)
        key = (subtree_mode, float(eps), bool(with_bias))
        if key != self._mode:
            _lib.check(self.lib.xfr_engine_set_mode(self._h, MODES[subtree_mode], float(eps), 1 if with_bias else 0))
            self._mode = key

    def tensor_shape(self, tid):
        c, h, w = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        _lib.check(self.lib.xfr_engine_tensor_shape(self._h, int(tid), ctypes.byref(c), ctypes.byref(h), ctypes.byref(w)))
        return (c.value, h.value, w.value)

    def memory(self):
        a, b = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(self.lib.xfr_engine_memory(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    # ------------------------------------------------------------------------------------------------
    def _prep(self, x):
        """-> (device tensor, fresh).  fresh: the tensor was produced here (host-to-device copy, cast, .contiguous()), i.e. it
        is still pending on the current stream; such an input must never be declared inputs_ready to the engine, whose
        pipelined forwards run on internal streams that do not wait for the caller's stream."""
        if x.dim() != 4 or tuple(x.shape[1:]) != tuple(self.program.in_shape):
            raise ValueError('expected input N x %s, got %s' % (self.program.in_shape, tuple(x.shape)))
        if x.shape[0] > self.max_batch:
            raise ValueError('batch %d exceeds the engine max_batch %d' % (x.shape[0], self.max_batch))
        y = x.detach().to(self.device, torch.float32).contiguous()
        fresh = (not x.is_cuda) or y.data_ptr() != x.data_ptr()
        return y, fresh

    def _declare_ready(self, ready):
        """Pipeline level 2: tell the engine whether the NEXT ebp / contrastive call may read x without waiting for the
        caller's stream (xfr_engine_set_inputs_ready; the engine consumes the promise with that call, so ebp_capture /
        ebp_firing / layerwise, which never declare anything, always take the safe ordering)."""
        if self._pipeline & 2:
            _lib.check(self.lib.xfr_engine_set_inputs_ready(self._h, 1 if ready else 0))

    def forward(self, x, tensor_id):
        x, _ = self._prep(x)
        c, h, w = self.tensor_shape(tensor_id)
        out = torch.empty((x.shape[0], c, h, w), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.xfr_forward(self._h, x.data_ptr(), x.shape[0], int(tensor_id), out.data_ptr(),
                                            _stream_ptr(self.device)))
        return out

    def ebp(self, x, seed_tensor, seed, want_mwp=False, want_pooled=True, inputs_ready=False):
        """seed: S x N x D.  Returns (mwp S x N x C1 x H1 x W1 or None, pooled S x N x H1 x W1 or None).
        inputs_ready (pipeline level 2 only): x is resident and valid on the device, see set_pipeline."""
        x, fresh = self._prep(x)
        self._declare_ready(inputs_ready and not fresh)
        n = x.shape[0]
        seed = seed.detach().to(self.device, torch.float32).contiguous()
        S = seed.shape[0]
        d = int(np.prod(self.tensor_shape(seed_tensor)))
        if seed.dim() != 3 or seed.shape[1] != n or seed.shape[2] != d:
            raise ValueError('seed must be S x %d x %d, got %s' % (n, d, tuple(seed.shape)))
        c1, h1, w1 = self.tensor_shape(1)
        mwp = torch.empty((S, n, c1, h1, w1), device=self.device) if want_mwp else None
        pooled = torch.empty((S, n, h1, w1), device=self.device) if want_pooled else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.xfr_ebp(self._h, x.data_ptr(), n, S, int(seed_tensor), seed.data_ptr(),
                                        mwp.data_ptr() if want_mwp else None,
                                        pooled.data_ptr() if want_pooled else None, _stream_ptr(self.device)))
        return mwp, pooled

    def contrastive(self, x, seed_tensor, seed, percentile=None, raw=False, inputs_ready=False):
        """seed: 2 x N x D (mate, non-mate).  Returns N x H1 x W1 saliency maps (raw=True: the contrastive MWP before
        _mwp_to_saliency)."""
        x, fresh = self._prep(x)
        self._declare_ready(inputs_ready and not fresh)
        n = x.shape[0]
        seed = seed.detach().to(self.device, torch.float32).contiguous()
        d = int(np.prod(self.tensor_shape(seed_tensor)))
        if tuple(seed.shape) != (2, n, d):
            raise ValueError('seed must be 2 x %d x %d, got %s' % (n, d, tuple(seed.shape)))
        c1, h1, w1 = self.tensor_shape(1)
        sal = torch.empty((n, h1, w1), device=self.device)
        pct = -1.0 if percentile is None else float(percentile)
        with torch.cuda.device(self.device):
            fn = self.lib.xfr_contrastive_raw if raw else self.lib.xfr_contrastive
            _lib.check(fn(self._h, x.data_ptr(), n, int(seed_tensor), seed.data_ptr(), pct, sal.data_ptr(), _stream_ptr(self.device)))
        return sal

    def triplet_contrastive(self, probes, gallery, encode_tensor, scale=1.0 / 2500.0, percentile=None, inputs_ready=False):
        """probes N x C x H x W, gallery 2N x C x H x W (mates then non-mates) -> N x H1 x W1 saliency maps.
        inputs_ready=True: both tensors are already valid on the device (not pending on the current stream) and will not be
        modified or freed until the result has been consumed -- required for cross-call pipelining (set_pipeline)."""
        probes, fresh = self._prep(probes)
        n = probes.shape[0]
        g_in = gallery
        gallery = gallery.detach().to(self.device, torch.float32).contiguous()
        if fresh or (not g_in.is_cuda) or gallery.data_ptr() != g_in.data_ptr():
            inputs_ready = False          # a copy made here is still pending on the current stream
        if tuple(gallery.shape) != (2 * n,) + tuple(self.program.in_shape):
            raise ValueError('gallery must be %d x %s, got %s' % (2 * n, self.program.in_shape, tuple(gallery.shape)))
        c1, h1, w1 = self.tensor_shape(1)
        sal = torch.empty((n, h1, w1), device=self.device)
        pct = -1.0 if percentile is None else float(percentile)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.xfr_triplet_contrastive(self._h, probes.data_ptr(), gallery.data_ptr(), n, int(encode_tensor),
                                                        float(scale), pct, sal.data_ptr(), _stream_ptr(self.device),
                                                        1 if inputs_ready else 0))
        return sal

    # -- uint8 inputs (include/xfr_amd.h: xfr_forward_u8 / xfr_triplet_contrastive_u8) ---------------------------------
    def set_u8_preprocess(self, kind, channels=3, mean=None, weight=None):
        """kind 'sub_mean' (mean per channel; ResNet-101 / ResNet-50-128d) or 'luminance' (weights per channel of value / 255; Light-CNN)."""
        p = _lib.U8Preprocess()
        p.kind = {'sub_mean': 0, 'luminance': 1}[kind]
        p.channels = int(channels)
        for i, v in enumerate(mean or ()):
            p.mean[i] = float(v)
        for i, v in enumerate(weight or ()):
            p.weight[i] = float(v)
        _lib.check(self.lib.xfr_engine_set_u8_preprocess(self._h, ctypes.byref(p)))
        self._u8_channels = int(channels)
        self.options['u8_preprocess'] = (kind, int(channels), tuple(mean or ()), tuple(weight or ()))

    def _prep_u8(self, x):
        """uint8 N x H x W x C (as decoded) -> (device tensor, fresh); see _prep."""
        c, h, w = self.program.in_shape
        if x.dtype != torch.uint8 or x.dim() != 4 or tuple(x.shape[1:]) != (h, w, getattr(self, '_u8_channels', 3)):
            raise ValueError('expected uint8 images N x %d x %d x %d, got %s %s' % (h, w, getattr(self, '_u8_channels', 3), x.dtype, tuple(x.shape)))
        if x.shape[0] > self.max_batch:
            raise ValueError('batch %d exceeds the engine max_batch %d' % (x.shape[0], self.max_batch))
        y = x.detach().to(self.device, non_blocking=True).contiguous()
        return y, (not x.is_cuda) or y.data_ptr() != x.data_ptr()

    def preprocess_u8(self, x):
        """The fp32 network input (N x C x H x W) the uint8 path builds: parity hook for the reference's preprocess functions."""
        x, _ = self._prep_u8(x)
        out = torch.empty((x.shape[0],) + tuple(self.program.in_shape), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.xfr_debug_u8_preprocess(self._h, x.data_ptr(), x.shape[0], out.data_ptr(), _stream_ptr(self.device)))
        return out

    def forward_u8(self, x, tensor_id):
        x, _ = self._prep_u8(x)
        c, h, w = self.tensor_shape(tensor_id)
        out = torch.empty((x.shape[0], c, h, w), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.xfr_forward_u8(self._h, x.data_ptr(), x.shape[0], int(tensor_id), out.data_ptr(), _stream_ptr(self.device)))
        return out

    def triplet_contrastive_u8(self, probes, gallery, encode_tensor, scale=1.0 / 2500.0, percentile=None, inputs_ready=False):
        """triplet_contrastive on uint8 N x H x W x C probes and 2N gallery images (mates then non-mates)."""
        probes, f1 = self._prep_u8(probes)
        gallery, f2 = self._prep_u8(gallery)
        n = probes.shape[0]
        if gallery.shape[0] != 2 * n:
            raise ValueError('gallery must hold %d images, got %d' % (2 * n, gallery.shape[0]))
        c1, h1, w1 = self.tensor_shape(1)
        sal = torch.empty((n, h1, w1), device=self.device)
        pct = -1.0 if percentile is None else float(percentile)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.xfr_triplet_contrastive_u8(self._h, probes.data_ptr(), gallery.data_ptr(), n, int(encode_tensor), float(scale), pct,
                                                           sal.data_ptr(), _stream_ptr(self.device), 1 if (inputs_ready and not f1 and not f2) else 0))
        return sal

    def triplet_contrastive_u8_host(self, probes, gallery, encode_tensor, scale=1.0 / 2500.0, percentile=None):
        """triplet_contrastive on uint8 images in HOST memory (N x H x W x C probes, 2N gallery images; pinned tensors preferably): the engine copies them
        on its own stream into its own staging buffers, so with set_pipeline on the copy and the forwards of this call overlap the previous call's sweep
        (include/xfr_amd.h: xfr_triplet_contrastive_u8_host).  The copy is asynchronous: call wait_inputs_copied() before rewriting the host tensors."""
        c, h, w = self.program.in_shape
        ch = getattr(self, '_u8_channels', 3)
        for x in (probes, gallery):
            if x.is_cuda or x.dtype != torch.uint8 or x.dim() != 4 or tuple(x.shape[1:]) != (h, w, ch) or not x.is_contiguous():
                raise ValueError('expected contiguous uint8 HOST images N x %d x %d x %d, got %s %s on %s' % (h, w, ch, x.dtype, tuple(x.shape), x.device))
        n = probes.shape[0]
        if gallery.shape[0] != 2 * n:
            raise ValueError('gallery must hold %d images, got %d' % (2 * n, gallery.shape[0]))
        if 2 * n > self.max_batch:
            raise ValueError('batch %d exceeds the engine max_batch %d' % (2 * n, self.max_batch))
        c1, h1, w1 = self.tensor_shape(1)
        sal = torch.empty((n, h1, w1), device=self.device)
        pct = -1.0 if percentile is None else float(percentile)
        self._host_inputs = (probes, gallery)        # keep them alive until the next call replaces them (the copy is asynchronous)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.xfr_triplet_contrastive_u8_host(self._h, probes.data_ptr(), gallery.data_ptr(), n, int(encode_tensor), float(scale), pct,
                                                                sal.data_ptr(), _stream_ptr(self.device)))
        return sal

    def wait_inputs_copied(self):
        """Block until the host-to-device copies of the last triplet_contrastive_u8_host call are done (its host tensors may then be rewritten)."""
        _lib.check(self.lib.xfr_engine_wait_inputs_copied(self._h))

    def set_pipeline(self, on):
        """Let the forward of triplet call i+1 overlap the backward of call i (see include/xfr_amd.h for the contract)."""
        _lib.check(self.lib.xfr_engine_set_pipeline(self._h, int(on)))     # 0 off, 1 triplet calls, 2 every run call; | 4: three forward slots
        self._pipeline = int(on)
        self.options['pipeline'] = int(on)

    def set_epilogue_fusion(self, on):
        """Hook chains / BatchNorm+add+ReLU inside the GEMM epilogue (default on) or as their own launches."""
        level = int(on) if not isinstance(on, bool) else (3 if on else 0)      # 0 off, 3 default (1: without the probe forward's BatchNorm / ReLU)
        _lib.check(self.lib.xfr_engine_set_epilogue_fusion(self._h, level))
        self.options['epilogue_fusion'] = level

    def hold_forward(self, on):
        """Consecutive calls on the same input tensor share one forward pass while held (include/xfr_amd.h)."""
        # re-entrant: nested holders (run_jobs_batched around weighted_subtree_ebp) keep one group open
        self._hold_depth = max(0, getattr(self, '_hold_depth', 0) + (1 if on else -1))
        if (on and self._hold_depth == 1) or (not on and self._hold_depth == 0):
            _lib.check(self.lib.xfr_engine_hold_forward(self._h, int(bool(on))))

    def set_tail_balance(self, on):
        """GEMM tail balancing (default on); off = batch-invariant fp32 arithmetic (include/xfr_amd.h)."""
        _lib.check(self.lib.xfr_engine_set_tail_balance(self._h, int(bool(on))))
        self.options['tail_balance'] = bool(on)

    def set_forward_split(self, on):
        """Forward-only batches of >= 32 images as two half batches on the internal streams (default on; include/xfr_amd.h)."""
        _lib.check(self.lib.xfr_engine_set_forward_split(self._h, int(bool(on))))
        self.options['forward_split'] = bool(on)

    def set_lean(self, on):
        """The lean schedule of un-observed sweeps (default on): stored hook quotients / one-bit gates instead of the literal hook operands;
        off = the literal expressions of whitebox.py:388-428 everywhere (include/xfr_amd.h)."""
        _lib.check(self.lib.xfr_engine_set_lean(self._h, int(bool(on))))
        self.options['lean'] = bool(on)

    def set_split_gemm(self, mode):
        """bf16x6 GEMMs for the deep-K stride-1 convolutions (include/xfr_amd.h, conv_gemm_split.hip K17): 0 / False off, 1 / True the forward
        convolutions, 2 the sweep's backward-data GEMMs, 3 both (the default); + 4: whatever the launch's grid (by default launches of fewer than 128
        tiles stay on the fp32 kernels)."""
        mode = int(mode)
        _lib.check(self.lib.xfr_engine_set_split_gemm(self._h, mode))
        self.options['split_gemm'] = mode

    def split_gemm_launches(self):
        """Launches of the bf16x6 kernel so far (process-wide)."""
        n = ctypes.c_int64(0)
        _lib.check(self.lib.xfr_engine_split_gemm_stats(self._h, ctypes.byref(n)))
        return int(n.value)

    def lean_launches(self):
        """Convolution launches of this engine that took the lean (dual-accumulator) form so far."""
        n = ctypes.c_int64(0)
        _lib.check(self.lib.xfr_engine_lean_stats(self._h, ctypes.byref(n)))
        return int(n.value)

    def apply_options(self, options):
        """Re-apply switches recorded by another Engine handle (WhiteboxNetwork.engine rebuilds engines that are too small)."""
        if 'tail_balance' in options:
            self.set_tail_balance(options['tail_balance'])
        if 'forward_split' in options:
            self.set_forward_split(options['forward_split'])
        if 'lean' in options:
            self.set_lean(options['lean'])
        if 'split_gemm' in options:
            self.set_split_gemm(options['split_gemm'])
        if 'u8_preprocess' in options:
            k, c, m, w = options['u8_preprocess']
            self.set_u8_preprocess(k, c, m, w)
        if 'epilogue_fusion' in options:
            self.set_epilogue_fusion(options['epilogue_fusion'])
        if options.get('pipeline'):
            self.set_pipeline(options['pipeline'])

    def mwp_to_saliency(self, pooled):
        pooled = pooled.detach().to(self.device, torch.float32).contiguous()
        n, h, w = pooled.shape
        out = torch.empty_like(pooled)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.xfr_mwp_to_saliency(self._h, pooled.data_ptr(), n, h, w, out.data_ptr(),
                                                    _stream_ptr(self.device)))
        return out

    # -- "next" row: layerwise / weighted-subtree EBP ---------------------------------------------------
    def firing_count(self, seed_tensor):
        n = ctypes.c_int32()
        _lib.check(self.lib.xfr_firing_count(self._h, int(seed_tensor), ctypes.byref(n)))
        return n.value

    def firing_names(self, seed_tensor):
        """Class names of the hooked modules in firing order (Whitebox.P_layername without the argument lists)."""
        nf = self.firing_count(seed_tensor)
        kinds = (ctypes.c_int32 * max(nf, 1))()
        _lib.check(self.lib.xfr_firing_kinds(self._h, int(seed_tensor), kinds, nf))
        return [LAYER_NAMES[OpKind(k)] for k in list(kinds)[:nf]]

    def subtree_weights(self, x, seed_tensor, seed, gate_ge0=True):
        """seed 2 x N x D -> (w [n_firings, N] float32, idx [n_firings, N] int32)."""
        x, _ = self._prep(x)
        n = x.shape[0]
        seed = seed.detach().to(self.device, torch.float32).contiguous()
        nf = self.firing_count(seed_tensor)
        w = (ctypes.c_float * (nf * n))()
        idx = (ctypes.c_int32 * (nf * n))()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.xfr_subtree_weights(self._h, x.data_ptr(), n, int(seed_tensor), seed.data_ptr(), 1 if gate_ge0 else 0,
                                                    w, idx, nf * n, _stream_ptr(self.device)))
        return (np.frombuffer(w, dtype=np.float32).reshape(nf, n).copy(), np.frombuffer(idx, dtype=np.int32).reshape(nf, n).copy())

    def ebp_capture(self, x, seed_tensor, seed, elems):
        """N images, seed 1 x N x D; elems[k][b] = flattened (c,h,w) element of firing k for image b (a flat list of n_firings
        ints is accepted for N = 1) -> float32 [n_firings, N] of P[k][b].flatten()[elems[k][b]]."""
        x, _ = self._prep(x)
        n = x.shape[0]
        seed = seed.detach().to(self.device, torch.float32).contiguous()
        nf = self.firing_count(seed_tensor)
        el_np = np.ascontiguousarray(np.asarray(elems, dtype=np.int32).reshape(nf, n))
        out = np.zeros((nf, n), dtype=np.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.xfr_ebp_capture(self._h, x.data_ptr(), n, int(seed_tensor), seed.data_ptr(),
                                                el_np.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                                out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), nf, _stream_ptr(self.device)))
        return out if n > 1 else out[:, 0]

    def layerwise(self, x, seed_tensor, firings, elems=None, vals=None, dense_prior=None):
        """Layerwise sweeps of N images sharing their forwards -> pooled P[-2].
        One image: `firings` is a list of J firings (any order) -> J x H1 x W1, in the order given.
        N images: `firings` / `elems` / `vals` are J x N arrays (firing < 0: idle sweep) -> J x N x H1 x W1; hand the sweeps of
        every image over in ascending firing order so that sweep j joins the backward pass where its first prior fires."""
        x, _ = self._prep(x)
        n = x.shape[0]
        f_np = np.asarray(firings, dtype=np.int32)
        if f_np.ndim == 1 and n == 1 and dense_prior is None:
            order = np.argsort(f_np, kind='stable')
            out = self.layerwise(x, seed_tensor, f_np[order].reshape(-1, 1), np.asarray(elems, dtype=np.int32)[order].reshape(-1, 1),
                                 np.asarray(vals, dtype=np.float32)[order].reshape(-1, 1))
            inv = np.empty_like(order)
            inv[order] = np.arange(len(order))
            return out[torch.as_tensor(inv, device=out.device), 0]
        f_np = np.ascontiguousarray(f_np.reshape(-1, n))
        J = f_np.shape[0]
        c1, h1, w1 = self.tensor_shape(1)
        out = torch.empty((J, n, h1, w1), device=self.device)
        el = va = dp = None
        if dense_prior is not None:
            dense_prior = dense_prior.detach().to(self.device, torch.float32).contiguous()
            dp = dense_prior.data_ptr()
        else:
            el_np = np.ascontiguousarray(np.asarray(elems, dtype=np.int32).reshape(J, n))
            va_np = np.ascontiguousarray(np.asarray(vals, dtype=np.float32).reshape(J, n))
            el = el_np.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
            va = va_np.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        with torch.cuda.device(self.device):
            _lib.check(self.lib.xfr_layerwise_ebp(self._h, x.data_ptr(), n, J, int(seed_tensor),
                                                  f_np.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), el, va, dp, out.data_ptr(),
                                                  _stream_ptr(self.device)))
        return out if (n > 1 or dense_prior is None) else out[:, 0]

    def ebp_firing(self, x, seed_tensor, seed, firing):
        """Whitebox.P[firing] of a standard sweep: N x C x H x W."""
        x, _ = self._prep(x)
        n = x.shape[0]
        seed = seed.detach().to(self.device, torch.float32).contiguous()
        c, h, w = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.xfr_ebp_store_firing(self._h, x.data_ptr(), n, int(seed_tensor), seed.data_ptr(), int(firing), None,
                                                     ctypes.byref(c), ctypes.byref(h), ctypes.byref(w), _stream_ptr(self.device)))
            out = torch.empty((n, c.value, h.value, w.value), device=self.device)
            _lib.check(self.lib.xfr_ebp_store_firing(self._h, x.data_ptr(), n, int(seed_tensor), seed.data_ptr(), int(firing),
                                                     out.data_ptr(), None, None, None, _stream_ptr(self.device)))
        return out

    # ------------------------------------------------------------------------------------------------
    def set_trace(self, on):
        _lib.check(self.lib.xfr_engine_set_trace(self._h, 1 if on else 0))

    def get_trace(self):
        """(sums [n_firings, S*N] float64, names [n_firings]) of the last ebp call, reference firing order."""
        nf = ctypes.c_int32()
        _lib.check(self.lib.xfr_engine_trace_size(self._h, ctypes.byref(nf)))
        cap = nf.value * 2 * self.max_batch
        sums = (ctypes.c_double * max(cap, 1))()
        kinds = (ctypes.c_int32 * max(nf.value, 1))()
        _lib.check(self.lib.xfr_engine_get_trace(self._h, sums, kinds, cap))
        arr = np.frombuffer(sums, dtype=np.float64)
        names = [LAYER_NAMES[OpKind(k)] for k in list(kinds)[:nf.value]]
        return arr, names, nf.value

    def set_profile(self, on):
        _lib.check(self.lib.xfr_engine_set_profile(self._h, 1 if on else 0))

    def profile_csv(self, path):
        """Append one CSV record per GEMM launch to `path` while profiling is on (None: stop)."""
        _lib.check(self.lib.xfr_engine_profile_csv(self._h, path.encode() if path else None))

    def get_profile_by_kernel(self):
        """The last profiled run's GEMM launches by the kernel family that ran them: {'fp32': (ms, launches, flops), 'bf16x6': (...)}."""
        ms, n, fl = (ctypes.c_double * 2)(), (ctypes.c_int64 * 2)(), (ctypes.c_double * 2)()
        _lib.check(self.lib.xfr_engine_get_profile_by_kernel(self._h, ms, n, fl))
        return {'fp32': (ms[0], n[0], fl[0]), 'bf16x6': (ms[1], n[1], fl[1])}

    def get_profile(self):
        ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        _lib.check(self.lib.xfr_engine_get_profile(self._h, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)))
        return ms.value, n.value, fl.value

