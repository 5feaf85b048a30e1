"""On-disk saliency format + resume (xfr_amd/saliency_io.py; reference python/xfr/show.py:196-232,131-137,46-129)."""
import os

import numpy as np
import pytest

from xfr_amd import saliency_io as SIO


def test_method_names_match_generator_stems():
    # generate_whitebox_saliency.py:304-309,340-356,383-391
    assert SIO.method_name('meanEBP', 'affineonly_with_prior', 6, 'cuda') == 'meanEBP_mode=awp_v06_cuda'
    assert SIO.method_name('contrastive', 'norelu', 6, 'cpu') == 'contrastive_triplet_ebp_mode=norelu_v06_cpu'
    assert SIO.method_name('contrastive', 'norelu', 6, 'cuda', truncate_percent=20) == \
        'trunc_contrastive_triplet_ebp_mode=norelu_v06_pct20_cuda'
    assert SIO.method_name('weighted-subtree', 'affineonly_with_prior', 6, 'cuda', topk=32, mode_weighted='norelu') == \
        'weighted_subtree_triplet_ebp_mode=awp,norelu_v06_top32_cuda'
    with pytest.raises(RuntimeError):
        SIO.method_name('rise', 'norelu')
    assert SIO.saliency_paths('/o', '00012', 'm') == ('/o/00012-m-saliency-overlay.png', '/o/00012-m-saliency.npz')


def test_jet_matches_matplotlib_table():
    # values of matplotlib.cm.jet at 0, 0.5, 1 and at a knot
    np.testing.assert_allclose(SIO.jet(0.0), [0.0, 0.0, 0.5], atol=1e-12)
    np.testing.assert_allclose(SIO.jet(1.0), [0.5, 0.0, 0.0], atol=1e-12)
    np.testing.assert_allclose(SIO.jet(0.5), [0.4901960784, 1.0, 0.4775458570], atol=1e-6)
    np.testing.assert_allclose(SIO.jet(2.0), SIO.jet(1.0))
    assert SIO.jet(np.zeros((3, 4))).shape == (3, 4, 3)


def test_process_saliency_resize_semantics():
    rng = np.random.default_rng(0)
    sm = rng.random((112, 112))
    img = np.zeros((224, 224, 3))
    out = SIO.processSaliency(img, sm)
    assert out.shape == (224, 224) and out.min() >= 0.0 and out.max() <= 1.0 + 1e-12
    # same size: normalisation only (show.py:133-134)
    same = SIO.processSaliency(np.zeros((112, 112, 3)), sm)
    np.testing.assert_allclose(same, (sm - sm.min()) / ((sm - sm.min()).max() + 1e-9), atol=1e-12)
    # a cubic spline reproduces a linear ramp; pixel centres map as (i + 0.5) / zoom - 0.5 (grid_mode)
    ramp = np.tile(np.arange(32, dtype=np.float64), (32, 1))
    up = SIO._resize_cubic(ramp, (64, 64))
    i = np.arange(16, 48)
    np.testing.assert_allclose(up[32, i], (i + 0.5) / 2 - 0.5, atol=1e-3)   # the zero border decays into the interior
    # shrinking pre-smooths and stays inside the input range
    down = SIO._resize_cubic(sm, (56, 56))
    assert down.shape == (56, 56) and down.min() >= 0.0 and down.max() <= sm.max()


def test_jet_equals_matplotlib_everywhere():
    """The overlay colormap against the library the reference uses (show.py:13,94: matplotlib's 'jet'), on a dense grid."""
    mpl = pytest.importorskip('matplotlib')
    import matplotlib.cm  # noqa: F401
    try:
        cmap = mpl.colormaps['jet']
    except AttributeError:
        cmap = mpl.cm.get_cmap('jet')
    x = np.linspace(0.0, 1.0, 4097)
    np.testing.assert_allclose(SIO.jet(x), cmap(x)[:, :3], atol=1e-12)


def test_resize_edge_cases():
    """processSaliency's resize (show.py:136) on the shapes the generator meets besides 112 -> 224: non-square probes, shrinking
    (IJB-C crops smaller than the map), identity, a constant map; and resize_linear (whitebox.py:802) on non-224 inputs."""
    rng = np.random.default_rng(3)
    sm = rng.random((112, 112))
    for shape in ((160, 128), (96, 300), (56, 56), (33, 71)):
        out = SIO._resize_cubic(sm, shape)
        assert out.shape == shape and np.isfinite(out).all()
        assert out.min() >= min(sm.min(), 0.0) - 1e-12 and out.max() <= sm.max() + 1e-12          # clip=True semantics
    np.testing.assert_array_equal(SIO._resize_cubic(sm, (112, 112)), sm)
    c = SIO._resize_cubic(np.full((20, 30), 0.7), (40, 45))
    np.testing.assert_allclose(c[8:-8, 8:-8], 0.7, atol=2e-3)                                      # away from the zero border
    # mean-preserving away from the border: the up-sampled map integrates like the original
    up = SIO._resize_cubic(sm, (224, 224))
    assert abs(up[16:-16, 16:-16].mean() - sm[8:-8, 8:-8].mean()) < 2e-3
    img = rng.random((100, 80, 3))
    lin = SIO.resize_linear(img, (224, 224))
    assert lin.shape == (224, 224, 3) and lin.min() >= img.min() - 1e-12 and lin.max() <= img.max() + 1e-12
    assert SIO.resize_linear(img, (100, 80)) is img                                                # same size: the identity
    small = SIO.resize_linear(rng.random((448, 448, 3)), (224, 224))                                # shrinking pre-smooths
    assert small.shape == (224, 224, 3) and small.std() < 0.2


def test_blend_properties():
    img = np.full((64, 48, 3), 0.25)
    sm = np.zeros((16, 12)); sm[8, 6] = 1.0
    ov = SIO.blend_saliency_map(img, sm)
    assert ov.shape == (64, 48, 3) and ov.min() >= 0.0 and ov.max() <= 1.0
    np.testing.assert_allclose(ov[0, 0], img[0, 0] * (1 - 0.0) + 0.0, atol=1e-6)      # zero saliency: the image shows through
    assert np.abs(ov[34, 26] - img[34, 26]).max() > 0.3                                 # the hot spot is coloured
    np.testing.assert_allclose(SIO.blend_saliency_map(img, np.ones((4, 4))), img)       # constant map is suppressed


def test_create_save_smap_format_and_resume(tmp_path):
    out = str(tmp_path / 'subject_ID_7' / 'aligned')
    probe = (np.random.default_rng(1).random((160, 128, 3)) * 255).astype(np.uint8)
    calls = []

    def smap_fn():
        calls.append(1)
        m = np.random.default_rng(2).random((112, 112)).astype(np.float32)
        return m / m.sum()

    name = SIO.method_name('contrastive', 'affineonly_with_prior', 6, 'cuda')
    assert SIO.create_save_smap(name, out, False, smap_fn, '00003', probe) is True
    ov, npz = SIO.saliency_paths(out, '00003', name)
    assert os.path.exists(ov) and os.path.exists(npz) and not os.path.exists(npz + '.tmp.npz')
    with np.load(npz) as z:
        assert list(z.keys()) == ['saliency_map']
        sm = z['saliency_map']
    assert sm.shape == (160, 128) and sm.min() >= 0.0 and abs(sm.max() - 1.0) < 1e-3       # min-max normalised, probe-sized
    import PIL.Image
    assert PIL.Image.open(ov).size == (128, 160)
    # resume: nothing is recomputed while both files exist
    assert SIO.create_save_smap(name, out, False, smap_fn, '00003', probe) is False and len(calls) == 1
    # a missing npz (interrupted job) or overwrite=True recomputes
    os.remove(npz)
    assert SIO.create_save_smap(name, out, False, smap_fn, '00003', probe) is True and len(calls) == 2
    assert SIO.create_save_smap(name, out, True, smap_fn, '00003', probe) is True and len(calls) == 3
    np.testing.assert_array_equal(SIO.load_smap(out, '00003', name), sm)


def _bspline3(t):
    """Cubic B-spline basis, written out (support [-2, 2])."""
    t = np.abs(t)
    return np.where(t < 1, 2.0 / 3.0 - t * t + 0.5 * t ** 3, np.where(t < 2, (2.0 - t) ** 3 / 6.0, 0.0))


def _spline_resize_by_hand(img, out_shape):
    """Cubic-spline resampling derived from first principles, independent of scipy.ndimage: (1) the B-spline coefficients c of
    every row / column solve the tridiagonal system (c[k-1] + 4 c[k] + c[k+1]) / 6 = s[k] (dense solve, 'not-a-knot-free' ends:
    rows of the matrix truncated at the border, whose influence decays like (sqrt(3) - 2)^d, i.e. 4e-12 at 20 pixels); (2) output
    pixel i samples the spline at x = (i + 0.5) * n_in / n_out - 0.5 (pixel centres map to pixel centres: what skimage >= 0.19
    obtains from zoom(..., grid_mode=True))."""
    out = np.asarray(img, dtype=np.float64)
    for ax, n_out in enumerate(out_shape):
        n_in = out.shape[ax]
        A = (np.diag(np.full(n_in, 4.0)) + np.diag(np.ones(n_in - 1), 1) + np.diag(np.ones(n_in - 1), -1)) / 6.0
        c = np.linalg.solve(A, np.moveaxis(out, ax, 0).reshape(n_in, -1))
        x = (np.arange(n_out) + 0.5) * n_in / n_out - 0.5
        Wm = _bspline3(x[:, None] - np.arange(n_in)[None, :])
        res = (Wm @ c).reshape((n_out,) + tuple(np.delete(out.shape, ax)))
        out = np.moveaxis(res, 0, ax)
    return out


@pytest.mark.parametrize('out', [(224, 224), (160, 160), (160, 224)])
def test_cubic_resize_equals_the_hand_derived_spline_in_the_interior(out):
    """processSaliency's resize (show.py:136: skimage.transform.resize(order=3, mode='constant')) at the generator's sizes
    (112 -> 224) and a non-integer factor (112 -> 160): away from the border -- where the 'constant' padding has decayed -- the
    module's restatement equals cubic-spline resampling written out by hand.  Parity with skimage ITSELF stays unpinned (the
    package cannot be installed in this image); this pins the restatement to the mathematics skimage's release notes and source
    (v0.19.0, skimage/transform/_warps.py: resize -> scipy.ndimage.zoom(order=3, grid_mode=True)) describe."""
    rng = np.random.default_rng(3)
    sm = ndi_smooth(rng.random((112, 112)))
    got = SIO._resize_cubic(sm, out)
    want = _spline_resize_by_hand(sm, out)
    m = 48                                           # output pixels: 24 input pixels at the 2x zoom
    assert got.shape == out
    np.testing.assert_allclose(got[m:-m, m:-m], np.clip(want, min(sm.min(), 0), max(sm.max(), 0))[m:-m, m:-m], rtol=0, atol=1e-9)


def ndi_smooth(a):
    import scipy.ndimage as ndi
    return ndi.gaussian_filter(a, 3.0)


def test_cubic_resize_analytic_cases():
    """What cubic-spline resampling must do exactly: a constant stays constant and a linear ramp stays the same ramp in the
    interior (splines reproduce polynomials up to degree 3), the same size is the identity everywhere, and the zero padding of
    mode='constant' can only darken the border, never brighten it."""
    c = SIO._resize_cubic(np.full((112, 112), 0.37), (224, 224))
    np.testing.assert_allclose(c[40:-40, 40:-40], 0.37, atol=1e-12)
    assert c.max() <= 0.37 + 1e-12 and c[0, 0] < 0.37                      # zero border bleeds in, clipping keeps the range
    yy, xx = np.mgrid[0:112, 0:112].astype(np.float64)
    ramp = 0.25 * yy + 0.5 * xx + 3.0
    for out in ((224, 224), (160, 160)):
        up = SIO._resize_cubic(ramp, out)
        fy, fx = 112.0 / out[0], 112.0 / out[1]
        oy, ox = np.mgrid[0:out[0], 0:out[1]].astype(np.float64)
        want = 0.25 * ((oy + 0.5) * fy - 0.5) + 0.5 * ((ox + 0.5) * fx - 0.5) + 3.0
        m = int(30 / min(fy, fx))
        np.testing.assert_allclose(up[m:-m, m:-m], want[m:-m, m:-m], atol=1e-9)
    sm = np.random.default_rng(5).random((112, 112))
    np.testing.assert_array_equal(SIO._resize_cubic(sm, (112, 112)), sm)
