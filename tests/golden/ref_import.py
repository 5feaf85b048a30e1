"""Import the real stresearch/xfr reference (read-only at /root/reference) in THIS container.

Only used by tests/golden/make_golden.py (fixture generation) and by the optional
`-m "not gpu"` cross-check tests that skip when /root/reference is absent.  Nothing here
runs on the GPU box: the reference does not travel, only the arrays it produced do.

The reference imports four third-party modules that are absent from this image
(skimage, torchvision, imageio, six).  None of them contribute arithmetic to the hot path
except skimage.filters.gaussian, which *is* scipy.ndimage.gaussian_filter(mode='nearest',
truncate=4.0) for a 2-D float image; we register that exact call under the skimage name.
"""
import os
import sys
import types

REF_ROOT = os.environ.get('XFR_REFERENCE_ROOT', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'python', 'xfr'))


def _install_shims():
    import numpy as np
    import scipy.ndimage

    try:
        import skimage.filters  # noqa: F401
        have_sk = True
    except ImportError:
        have_sk = False
    if not have_sk:
        sk = types.ModuleType('skimage')
        sk.filters = types.ModuleType('skimage.filters')

        def gaussian(img, sigma):
            # skimage.filters.gaussian(image, sigma) for a float32 2-D image: mode='nearest',
            # truncate=4.0, preserve dtype
            return scipy.ndimage.gaussian_filter(img, sigma, mode='nearest', truncate=4.0)
        sk.filters.gaussian = gaussian
        sk.transform = types.ModuleType('skimage.transform')

        def resize(img, output_shape, **kw):
            # skimage.transform.resize to the SAME spatial size is the identity in every skimage version (zoom factor 1:
            # no anti-aliasing filter, interpolation at integer coordinates); anything else is not restated here, so the
            # fixtures only ever hand the reference 224x224 inputs (Whitebox.convert_from_numpy, whitebox.py:802)
            if tuple(img.shape[:2]) != tuple(output_shape[:2]):
                raise NotImplementedError('skimage.transform.resize shim: identity only (%s -> %s)' % (img.shape, output_shape))
            return img if img.dtype.char in 'df' else img.astype(float)
        sk.transform.resize = resize
        sk.morphology = types.ModuleType('skimage.morphology')
        sk.color = types.ModuleType('skimage.color')

        def rgb2gray(rgb):
            rgb = np.asarray(rgb)
            if rgb.ndim == 2:
                return rgb.astype(np.float64) / 255.0
            return (rgb[..., :3].astype(np.float64) / 255.0) @ np.array([0.2125, 0.7154, 0.0721])
        sk.color.rgb2gray = rgb2gray
        for name in ('skimage', 'skimage.filters', 'skimage.transform', 'skimage.morphology', 'skimage.color'):
            sys.modules[name] = {'skimage': sk, 'skimage.filters': sk.filters, 'skimage.transform': sk.transform,
                                 'skimage.morphology': sk.morphology, 'skimage.color': sk.color}[name]

    try:
        import torchvision.transforms  # noqa: F401
        have_tv = True
    except ImportError:
        have_tv = False
    if not have_tv:
        tv = types.ModuleType('torchvision')
        tr = types.ModuleType('torchvision.transforms')

        class _T(object):
            def __init__(self, *a, **k):
                self.a = a

            def __call__(self, x):
                return x

        class Lambda(_T):
            def __call__(self, x):
                return self.a[0](x)

        class Compose(_T):
            def __call__(self, x):
                for t in self.a[0]:
                    x = t(x)
                return x
        tr.Resize = _T
        tr.CenterCrop = _T
        tr.Lambda = Lambda
        tr.Compose = Compose
        tv.transforms = tr
        sys.modules['torchvision'] = tv
        sys.modules['torchvision.transforms'] = tr

    try:
        import imageio  # noqa: F401
    except ImportError:
        sys.modules['imageio'] = types.ModuleType('imageio')
    try:
        import six  # noqa: F401  (present in this image; pandas needs the real one)
    except ImportError:
        six = types.ModuleType('six')
        six.string_types = (str,)
        sys.modules['six'] = six


def load():
    """Returns a namespace with the reference modules (whitebox, resnet, lightcnn, resnet50_128)."""
    if not available():
        raise RuntimeError('reference not present at %s' % REF_ROOT)
    _install_shims()
    for p in (os.path.join(REF_ROOT, 'python'), os.path.join(REF_ROOT, 'models', 'resnet50_128_pytorch')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        import xfr.models.whitebox as whitebox
        import xfr.models.resnet as resnet
        import xfr.models.lightcnn as lightcnn
        import resnet50_128
    ns = types.SimpleNamespace(whitebox=whitebox, resnet=resnet, lightcnn=lightcnn, resnet50_128=resnet50_128)
    return ns
