"""xfr_amd.image_loader: the file / DataFrame intake of Whitebox.embeddings and preprocess_loader (xfr/utils.py:39-202).  Geometry cases are
worked out by hand from the reference's expressions (the reference module itself needs imageio, absent here)."""
import os

import numpy as np
import PIL.Image
import pytest

from xfr_amd import image_loader as IL


def _img(h, w, seed=0):
    return np.random.RandomState(seed).randint(0, 256, size=(h, w, 3)).astype(np.uint8)


def test_crop_box_expand_and_borders():
    # box 40x60 at (10, 20) in a 200 x 300 image: side = max(w, h) = 60, centre (30, 50) -> top 20, left 0 (clamped), right 60
    assert IL.crop_box((200, 300), 10, 20, 40, 60) == (20, 80, 0, 60)
    # a box larger than the image's smaller side: side = min(H, W); the crop is pushed back inside
    assert IL.crop_box((100, 300), 150, 10, 200, 80) == (0, 100, 200, 300)
    # hits the bottom / right border: bottom = H, top shifted up so that the side is kept
    assert IL.crop_box((120, 120), 80, 90, 40, 40) == (80, 120, 80, 120)
    assert IL.crop_box((120, 120), 100, 100, 40, 40) == (80, 120, 80, 120)
    # constrict variants use the SMALLER box side
    assert IL.crop_box((200, 300), 100, 50, 40, 60, 'constrict') == (60, 100, 100, 140)
    assert IL.crop_box((200, 300), 100, 50, 40, 60, 'constrict50') == (70, 90, 110, 130)
    # floats are rounded first (int(round(.)))
    assert IL.crop_box((200, 300), 9.6, 20.4, 40.2, 59.7) == (20, 80, 0, 60)


def test_crop_image_xywh_and_tblr():
    im = _img(200, 300)
    c, roi = IL.crop_image(im, crop_xywh=(10, 20, 40, 60))
    assert roi == (20, 80, 0, 60) and np.array_equal(c, im[20:80, 0:60])
    # crop_tblr: the reference derives w from the vertical extent and h from the horizontal one (utils.py:124-125)
    c2, roi2 = IL.crop_image(im, crop_tblr=(20, 80, 10, 50))
    assert roi2 == IL.crop_box(im.shape, 10, 20, 60, 40)
    with pytest.raises(ValueError):
        IL.crop_image(im)


def test_center_crop_square_identity_and_dtype():
    im = _img(224, 224, 1)
    assert np.array_equal(IL.center_crop(im), im)                      # 224 x 224: crop and resize are both the identity
    f = im.astype(float) / 255
    out = IL.center_crop(f, convert_uint8=False)
    assert out.dtype == f.dtype and np.array_equal(out, f)
    u = IL.center_crop(f)                                              # float in [0, 1] -> * 255 -> uint8 (truncation)
    assert u.dtype == np.uint8 and np.array_equal(u, (f * 255).astype(np.uint8))
    wide = _img(224, 300, 2)
    assert np.array_equal(IL.center_crop(wide), wide[:, 38:262])       # (300 - 224) // 2 = 38
    tall = _img(100, 60, 3)
    assert IL.center_crop(tall).shape == (224, 224, 3)                 # other sizes: shape only (resize parity unpinned)


def test_image_loader_files_arrays_and_tuples(tmp_path):
    a, b = _img(224, 224, 4), _img(224, 260, 5)
    fa, fb = str(tmp_path / 'a.png'), str(tmp_path / 'b.png')
    PIL.Image.fromarray(a).save(fa)
    PIL.Image.fromarray(b).save(fb)
    got = list(IL.image_loader([fa, fb]))
    assert len(got) == 2 and got[0].dtype == np.float64
    assert np.array_equal(got[0], a.astype(float) / 255)
    assert np.array_equal(got[1], (b.astype(float) / 255)[:, 18:242])
    arr = a.astype(float) / 255
    t = list(IL.image_loader([arr, fb], returnImageIndex=True, returnFileName=True))
    assert t[0][0] is arr and t[0][1:] == (0, None) and t[1][1:] == (1, fb)
    r = list(IL.image_loader([arr], repeats=3))
    assert [x[-1] for x in r] == [0, 1, 2] and all(x[0] is arr and len(x) == 2 for x in r)
    with pytest.raises(NotImplementedError):
        list(IL.image_loader([7]))
    with pytest.raises(AssertionError):
        list(IL.image_loader([np.zeros((4, 4))]))
    g = str(tmp_path / 'g.png')                                        # grey file through crop_example_no_name: repeated to three channels
    PIL.Image.fromarray(a[:, :, 0]).save(g)
    img, sid, fn, sid2 = IL.crop_example_no_name({'Filename': g, 'SubjectID': 's1'})
    assert img.shape == (224, 224, 3) and sid == sid2 == 's1' and fn == g and np.array_equal(img[:, :, 2], a[:, :, 0].astype(float) / 255)


def test_image_loader_dataframe(tmp_path):
    pd = pytest.importorskip('pandas')
    a = _img(200, 300, 6)
    fa = str(tmp_path / 'a.png')
    PIL.Image.fromarray(a).save(fa)
    df = pd.DataFrame([{'Filename': fa, 'SubjectID': 'x', 'XMin': 10.0, 'YMin': 20.0, 'Width': 40.0, 'Height': 60.0}])
    (img, idx, fn), = list(IL.image_loader(df, returnImageIndex=True, returnFileName=True))
    assert idx == 0 and fn == fa and np.array_equal(img, (a.astype(float) / 255)[20:80, 0:60])
    df2 = pd.DataFrame([{'Filename': fa, 'SubjectID': 'x'}])              # no face box: the whole image (KeyError branch, utils.py:172-173)
    img2, = list(IL.image_loader(df2))
    assert np.array_equal(img2, a.astype(float) / 255)
