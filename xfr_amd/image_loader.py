"""Host-side image intake of the Whitebox callers: what `Whitebox.embeddings` / `preprocess_loader` hand to `convert_from_numpy`
when they are given FILE NAMES or an inpainting-game DataFrame instead of arrays (reference: python/xfr/utils.py:39-109 image_loader,
:111-155 crop_image, :157-174 crop_example_no_name, :176-202 center_crop).

Only data formats: no arithmetic of the hot path lives here.  Files are decoded with PIL (the reference uses imageio, which is not in
this image: both hand back the decoder's uint8 H x W [x C] array for PNG / JPEG).  `center_crop`'s resize is `saliency_io.resize_linear`,
the restatement of skimage's order-1 resize used by `Whitebox.convert_from_numpy` -- the identity at 224 x 224, parity-unpinned elsewhere
(skimage is absent here)."""
import os

import numpy as np

from .saliency_io import resize_linear

NET_SIDE = 224          # utils.py:189 imgScale


def imread(fn):
    """Decoded pixels of an image file as the reference's imageio.imread (utils.py:88,164,179) hands them over: uint8, H x W (grey) or H x W x C.
    8-bit grey, RGB and RGBA files (PIL modes L / RGB / RGBA): the decoder's array, the same from PIL and imageio.  Palette files (mode P -- most small
    PNGs): imageio's pillow plugin expands them, to RGBA when the file carries transparency and to RGB otherwise, and so does this.  Anything else (LA,
    16-bit, CMYK ...) raises: imageio would hand back the decoder's native dtype / channel count there and a silent conversion would diverge."""
    import PIL.Image
    with PIL.Image.open(fn) as im:
        if im.mode == 'P':
            im = im.convert('RGBA' if 'transparency' in im.info else 'RGB')
        if im.mode not in ('L', 'RGB', 'RGBA'):
            raise ValueError('%s: unsupported image mode %r (8-bit grey / RGB / RGBA / palette files only)' % (fn, im.mode))
        return np.asarray(im).copy()


def crop_box(shape, x, y, w, h, roi_method='expand'):
    """(top, bottom, left, right) of the square crop utils.py:111-155 cuts around the box (x, y, w, h): centred on the box, side =
    the larger box side ('expand', never more than the image's smaller side) or a fraction of the smaller one ('constrict*'),
    pushed back inside the image where it would stick out."""
    x, y, w, h = (int(round(v)) for v in (x, y, w, h))
    H, W = int(shape[0]), int(shape[1])
    if roi_method == 'expand':
        side = min(max(w, h), min(H, W))
    else:
        scale = {'constrict': 1.0, 'constrict80': 0.8, 'constrict50': 0.5}[roi_method]
        side = int(min(w, h) * scale)
    cy, cx = y + h // 2, x + w // 2
    top, left = max(0, cy - side // 2), max(0, cx - side // 2)
    bottom, right = min(H, top + side), min(W, left + side)
    top, left = max(0, min(top, bottom - side)), max(0, min(left, right - side))
    return top, bottom, left, right


def crop_image(img, crop_xywh=None, crop_tblr=None, roi_method='expand'):
    """utils.py:111-155: (square crop, (top, bottom, left, right)).  `crop_tblr` = (top, bottom, left, right) of a box; the reference takes
    the box's width from the vertical extent and its height from the horizontal one (:124-125) and so does this."""
    if crop_tblr is not None:
        t, b, l, r = (int(round(v)) for v in crop_tblr)
        x, y, w, h = l, t, b - t, r - l
    elif crop_xywh is not None:
        x, y, w, h = crop_xywh
    else:
        raise ValueError('crop_image needs crop_xywh or crop_tblr')
    top, bottom, left, right = crop_box(img.shape, x, y, w, h, roi_method)
    return img[top:bottom, left:right, :], (top, bottom, left, right)


def center_crop(img, convert_uint8=True):
    """utils.py:176-202: the centred square of the shorter side, resized to 224 x 224, in the dtype it came in (after the optional uint8
    conversion: a float image with maximum <= 1 is scaled by 255 first)."""
    if isinstance(img, str):
        img = imread(img)
    if convert_uint8 and img.dtype != np.uint8:
        if img.max() <= 1:
            img = img * 255
        img = img.astype(np.uint8)
        assert img.max() > 1
    side = min(img.shape[:2])
    y0, x0 = (img.shape[0] - side) // 2, (img.shape[1] - side) // 2
    sq = img[y0:y0 + side, x0:x0 + side]
    return resize_linear(sq, (NET_SIDE, NET_SIDE)).astype(sq.dtype)


def crop_example_no_name(ex, data_root=''):
    """utils.py:157-174: one row of an inpainting-game DataFrame -> (float RGB image in [0, 1] cropped to the face box when the row has
    one, SubjectID, Filename, SubjectID)."""
    img = imread(os.path.join(data_root, ex['Filename'])).astype(float) / 255
    if img.ndim == 2:
        img = np.repeat(img[:, :, np.newaxis], 3, axis=2)
    try:
        img, _ = crop_image(img, crop_xywh=(ex['XMin'], ex['YMin'], ex['Width'], ex['Height']))
    except KeyError:
        pass
    return img, ex['SubjectID'], ex['Filename'], ex['SubjectID']


def image_loader(images, returnImageIndex=False, returnFileName=False, repeats=1):
    """utils.py:39-109: iterate displayable images (float H x W x 3) over a DataFrame (columns Filename, SubjectID [, XMin, YMin, Width,
    Height]), a sequence of file names (decoded, / 255, centre-cropped to 224 x 224) or of H x W x 3 arrays (passed through, fn None).
    Yields the bare image, or a tuple (image [, index] [, file name] [, repeat number]) as the reference does."""
    def rows():
        try:
            import pandas as pd
            is_frame = isinstance(images, pd.DataFrame)
        except ImportError:            # no pandas, no DataFrame to be given
            is_frame = False
        if is_frame:
            for i, (_, info) in enumerate(images.iterrows()):
                img, _, fn, _ = crop_example_no_name(info)
                assert img.max() <= 1.0 and img.min() >= 0.0
                yield i, img, fn
            return
        for i, img in enumerate(images):
            if isinstance(img, np.ndarray):
                assert img.ndim == 3 and img.shape[2] == 3
                yield i, img, None
            elif isinstance(img, str):
                yield i, center_crop(imread(img).astype(float) / 255, convert_uint8=False), img
            else:
                raise NotImplementedError('Unhandled type %s' % type(img))

    for i, img, fn in rows():
        ret = [img]
        if returnImageIndex:
            ret.append(i)
        if returnFileName:
            ret.append(fn)
        if repeats == 1:
            yield ret[0] if len(ret) == 1 else tuple(ret)
        else:
            for r in range(repeats):
                yield tuple(ret + [r])
