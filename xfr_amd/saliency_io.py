"""On-disk saliency format of the reference's inpainting-game generator, with its skip-if-exists resume.

Mirrors, by name and argument meaning:
    create_save_smap        python/xfr/show.py:196-232   (npz key `saliency_map`, overlay PNG, resume on both files)
    processSaliency         python/xfr/show.py:131-137   (min-shift, / (max + 1e-9), cubic resize to the probe size)
    blend_saliency_map      python/xfr/show.py:46-129    (jet colormap, gamma 0.8 alpha blend)
    shorten_subtree_mode, method_name
                            python/xfr/inpainting_game/generate_whitebox_saliency.py:216-219, 295-399 (file stems)

Host-side glue around the device maps (numpy/scipy/PIL); nothing here is on the EBP hot path.

Resize note -- parity unpinned, restatement stated precisely: the reference calls `skimage.transform.resize(attMap,
img.shape[:2], order=3, mode='constant')` (show.py:136) and skimage is not installed in this image, so there is nothing to
run it against.  This module restates scikit-image 0.19-0.22 (`skimage/transform/_warps.py: resize`): for a 2-D float image
that release computes `factors = in_shape / out_shape`; when any factor > 1 and anti_aliasing (default True) it pre-filters with
`ndi.gaussian_filter(image, (factors - 1) / 2, cval=0, mode='constant')`; then `ndi.zoom(image, 1 / factors, order=3,
mode='grid-constant' (_to_ndimage_mode('constant')), cval=0, grid_mode=True)`; then `_clip_warp_output`: clip to
[min(in.min, cval), max(in.max, cval)].  (skimage < 0.19 went through `warp`, whose border handling differs; README.md:37 of
the reference only asks for >= 0.17.2, so the reference itself is not pinned to one behaviour here.)  Source restated: scikit-image release tag v0.19.0,
`skimage/transform/_warps.py`, functions `resize` and `_clip_warp_output` (no network here: the tag is cited, not a commit hash
that could not be checked).  What IS pinned (tests/test_saliency_io.py): the restatement equals cubic-spline resampling written
out by hand -- tridiagonal B-spline coefficients, pixel-centre coordinate map -- to 1e-9 in the interior at 112 -> 224, 112 -> 160
and 112 -> 160 x 224; constants and linear ramps are reproduced exactly there; the same size is the identity; the zero padding
only darkens the border; clipping, shrinking and non-square outputs behave as stated; the jet table equals matplotlib's.
"""
import os

import numpy as np
import scipy.ndimage as ndi

__all__ = ['processSaliency', 'resize_linear', 'blend_saliency_map', 'create_save_smap', 'load_smap', 'shorten_subtree_mode', 'method_name',
           'saliency_paths', 'jet']


# ---- resize ------------------------------------------------------------------------------------------------------
def _resize_cubic(img, out_shape):
    img = np.asarray(img, dtype=np.float64)
    out_shape = tuple(int(v) for v in out_shape)
    if img.shape == out_shape:
        return img.copy()
    factors = np.divide(img.shape, out_shape)                # input / output per axis
    src = img
    if np.any(factors > 1):                                  # shrinking: gaussian pre-filter, sigma = (f - 1) / 2
        sigma = np.maximum(0, (factors - 1) / 2)
        src = ndi.gaussian_filter(img, sigma, cval=0, mode='constant')
    out = ndi.zoom(src, 1.0 / factors, order=3, mode='grid-constant', cval=0.0, grid_mode=True)
    if out.shape != out_shape:                               # zoom rounds the output size itself
        fixed = np.zeros(out_shape, dtype=out.dtype)
        h, w = min(out.shape[0], out_shape[0]), min(out.shape[1], out_shape[1])
        fixed[:h, :w] = out[:h, :w]
        out = fixed
    lo, hi = min(img.min(), 0.0), max(img.max(), 0.0)        # clip=True with mode='constant', cval=0
    return np.clip(out, lo, hi)


def resize_linear(img, out_shape):
    """skimage.transform.resize(img, out_shape, preserve_range=True) as Whitebox.convert_from_numpy calls it
    (whitebox.py:802: order 1, mode 'reflect', anti_aliasing on).  Same spatial size: the identity, in every skimage version
    (zoom factor 1 = no pre-filter and interpolation at integer coordinates) -- the case the golden fixtures pin.  Other sizes
    restate skimage >= 0.19 (`scipy.ndimage.zoom(order=1, mode='mirror', grid_mode=True)` after a gaussian pre-filter of
    sigma (f - 1) / 2 when shrinking) and are parity-unpinned like _resize_cubic."""
    out_shape = tuple(int(v) for v in out_shape[:2])
    if tuple(img.shape[:2]) == out_shape:
        return img if img.dtype.char in 'df' else img.astype(float)
    src = np.asarray(img, dtype=np.float64)
    factors = np.divide(src.shape[:2], out_shape)
    chans = src[..., None] if src.ndim == 2 else src
    outs = []
    for c in range(chans.shape[2]):
        a = chans[..., c]
        if np.any(factors > 1):
            a = ndi.gaussian_filter(a, np.maximum(0, (factors - 1) / 2), mode='mirror')
        z = ndi.zoom(a, 1.0 / factors, order=1, mode='mirror', grid_mode=True)
        fixed = np.zeros(out_shape, dtype=z.dtype)
        h, w = min(z.shape[0], out_shape[0]), min(z.shape[1], out_shape[1])
        fixed[:h, :w] = z[:h, :w]
        outs.append(np.clip(fixed, src.min(), src.max()))
    out = np.stack(outs, axis=2)
    return out[..., 0] if src.ndim == 2 else out


def processSaliency(img, attMap):
    """show.py:131-137: normalise to [0,1] and resize to the image's height x width."""
    attMap = np.asarray(attMap, dtype=np.float64)
    attMap = attMap - attMap.min()
    attMap = attMap / (attMap.max() + 1e-9)
    return _resize_cubic(attMap, np.asarray(img).shape[:2])


# ---- overlay -----------------------------------------------------------------------------------------------------
_JET = {   # matplotlib's 'jet' segment data (x, y): piecewise-linear, sampled on a 256-entry table like LinearSegmentedColormap
    'r': ((0.0, 0.0), (0.35, 0.0), (0.66, 1.0), (0.89, 1.0), (1.0, 0.5)),
    'g': ((0.0, 0.0), (0.125, 0.0), (0.375, 1.0), (0.64, 1.0), (0.91, 0.0), (1.0, 0.0)),
    'b': ((0.0, 0.5), (0.11, 1.0), (0.34, 1.0), (0.65, 0.0), (1.0, 0.0)),
}
_JET_LUT = None


def jet(x):
    """RGB of matplotlib's 'jet' for x in [0, 1] (256-entry lookup, values outside are clamped)."""
    global _JET_LUT
    if _JET_LUT is None:
        grid = np.linspace(0.0, 1.0, 256)
        _JET_LUT = np.stack([np.interp(grid, *zip(*_JET[c])) for c in 'rgb'], axis=1)
    x = np.asarray(x, dtype=np.float64)
    idx = np.clip((x * 256).astype(np.int64), 0, 255)
    idx = np.where(x >= 1.0, 255, idx)
    return _JET_LUT[idx]


def blend_saliency_map(image, smap, blur=False, blur_sigma=0.02, scale_factor=1.0, gamma=0.8):
    """show.py:46-129 for one image: image HxWx3 float in [0,1], smap any size; returns the HxWx3 overlay in [0,1]."""
    image = np.asarray(image, dtype=np.float64)
    att = np.array(smap, dtype=np.float64)
    att -= att.min()
    if att.max() <= 0:
        return image                                        # suppressed map: the image itself (show.py:110-111,127-128)
    att /= att.max()
    att = np.minimum(att, scale_factor) / scale_factor
    att = _resize_cubic(att, image.shape[:2])
    if blur:
        att = ndi.gaussian_filter(att, blur_sigma * max(image.shape[:2]), mode='nearest')
        att -= att.min()
        att /= att.max()
    alpha = (np.clip(att, 0.0, None) ** gamma)[..., None]
    return (1 - alpha) * image + alpha * jet(att)


# ---- file names and resume ---------------------------------------------------------------------------------------
def shorten_subtree_mode(ebp_subtree_mode):
    """generate_whitebox_saliency.py:216-219."""
    return 'awp' if ebp_subtree_mode == 'affineonly_with_prior' else ebp_subtree_mode


def method_name(kind, mode, ebp_ver=6, device_type='cuda', truncate_percent=None, topk=None, mode_weighted=None):
    """File stems of generate_whitebox_saliency.py:304-309, 340-356, 383-391."""
    m = shorten_subtree_mode(mode)
    if kind == 'meanEBP':
        return 'meanEBP_mode=%s_v%02d_%s' % (m, ebp_ver, device_type)
    if kind == 'contrastive':
        if truncate_percent is None:
            return 'contrastive_triplet_ebp_mode=%s_v%02d_%s' % (m, ebp_ver, device_type)
        return 'trunc_contrastive_triplet_ebp_mode=%s_v%02d_pct%d_%s' % (m, ebp_ver, truncate_percent, device_type)
    if kind == 'weighted-subtree':
        return 'weighted_subtree_triplet_ebp_mode=%s,%s_v%02d_top%d_%s' % (m, shorten_subtree_mode(mode_weighted), ebp_ver,
                                                                        topk, device_type)
    raise RuntimeError("Unknown method type %s (valid types: 'meanEBP', 'contrastive', 'weighted-subtree')" % kind)


def saliency_paths(output_dir, mask_id, method):
    """show.py:199-207."""
    return ('{}/{}-{}-saliency-overlay.png'.format(output_dir, mask_id, method),
            '{}/{}-{}-saliency.npz'.format(output_dir, mask_id, method))


def create_save_smap(method, output_dir, overwrite, smap_fn, mask_id, probe_im, probe_info=None, mask_im=None, verbose=False):
    """show.py:196-232.  `smap_fn()` is only called when the overlay or the npz is missing (or `overwrite`): that is the
    generator's resume.  Writes `{mask_id}-{method}-saliency.npz` (key `saliency_map`: float, probe-sized, in [0,1]) and the
    overlay PNG.  Returns True if the map was (re)computed."""
    overlay_filename, npz_filename = saliency_paths(output_dir, mask_id, method)
    if not overwrite and os.path.exists(overlay_filename) and os.path.exists(npz_filename):
        return False
    smap = np.asarray(smap_fn()).astype(np.float32)
    smap = smap - smap.min()
    smap = smap / smap.sum()
    probe_im = np.asarray(probe_im)
    img = probe_im.astype(np.float64) / (255.0 if probe_im.dtype == np.uint8 else 1.0)
    smap = processSaliency(img, smap)
    overlay = blend_saliency_map(img, smap)
    os.makedirs(output_dir, exist_ok=True)
    import PIL.Image
    # the npz goes last and atomically: an interrupted job leaves at most a stale overlay, never a half-written map
    PIL.Image.fromarray((np.clip(overlay, 0, 1) * 255).astype(np.uint8)).save(overlay_filename)
    tmp = npz_filename + '.tmp.npz'
    np.savez_compressed(tmp, saliency_map=smap)
    os.replace(tmp, npz_filename)
    if verbose:
        print('Created:\n %s\n' % overlay_filename)
    return True


def load_smap(output_dir, mask_id, method):
    """The reader side (plot_inpainting_game.py:228,258): the stored `saliency_map`."""
    return np.load(saliency_paths(output_dir, mask_id, method)[1])['saliency_map']
