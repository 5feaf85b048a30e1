"""Light-CNN-29 v2: mirror of python/xfr/models/lightcnn.py (v2 only) for the EBP hot path."""
import numpy as np
import torch

from ..program import Program
from ._backbone import Backbone


def rgb2gray(rgb):
    """skimage.color.rgb2gray for uint8 RGB -> float64 luminance in [0,1] (lightcnn.py:17,24)."""
    rgb = np.asarray(rgb)
    if rgb.ndim == 2:
        return rgb.astype(np.float64) / 255.0
    return (rgb[..., :3].astype(np.float64) / 255.0) @ np.array([0.2125, 0.7154, 0.0721])


def prepare_lightCNN_image(img):
    """lightcnn.py:19-25"""
    img_gray = rgb2gray(np.array(img))
    return torch.from_numpy(img_gray).float().unsqueeze(0).unsqueeze(0)


def lightcnn_preprocess():
    """Resize(144) -> CenterCrop(128) -> gray (lightcnn.py:27-31), with PIL instead of torchvision."""
    import PIL.Image

    def f(im):
        w, h = im.size
        if w <= h:
            nw, nh = 144, int(144 * h / w)
        else:
            nw, nh = int(144 * w / h), 144
        im = im.resize((nw, nh), PIL.Image.BILINEAR)
        left = int(round((nw - 128) / 2.0))
        top = int(round((nh - 128) / 2.0))
        im = im.crop((left, top, left + 128, top + 128))
        return prepare_lightCNN_image(im)
    return f


class network_29layers_v2(Backbone):
    """lightcnn.py:216-275 with block=resblock, layers=[1,2,3,4]."""
    arch = 'lightcnn29v2'
    in_shape = (1, 128, 128)

    def __init__(self, layers=(1, 2, 3, 4), num_classes=79077):
        super(network_29layers_v2, self).__init__()
        self.layers = tuple(layers)
        self.num_classes = int(num_classes)
        self.fc2_hooked = True
        self.init_parameters()

    def _mfms(self):
        """(prefix, cin, cout, k, pad) of every mfm in forward order, with structure tags."""
        L = self.layers
        seq = [('mfm', 'conv1', 1, 48, 5, 2), ('pool',)]
        seq += [('res', 'block1.%d' % i, 48) for i in range(L[0])]
        seq += [('group', 'group1', 48, 96), ('pool',)]
        seq += [('res', 'block2.%d' % i, 96) for i in range(L[1])]
        seq += [('group', 'group2', 96, 192), ('pool',)]
        seq += [('res', 'block3.%d' % i, 192) for i in range(L[2])]
        seq += [('group', 'group3', 192, 128)]
        seq += [('res', 'block4.%d' % i, 128) for i in range(L[3])]
        seq += [('group', 'group4', 128, 128), ('pool',)]
        return seq

    def param_specs(self):
        specs = []

        def mfm(p, cin, cout, k):
            specs.append((p + '.filter.weight', (2 * cout, cin, k, k), 'conv_w'))
            specs.append((p + '.filter.bias', (2 * cout,), 'conv_b'))
        for item in self._mfms():
            if item[0] == 'mfm':
                mfm(item[1], item[2], item[3], item[4])
            elif item[0] == 'res':
                mfm(item[1] + '.conv1', item[2], item[2], 3)
                mfm(item[1] + '.conv2', item[2], item[2], 3)
            elif item[0] == 'group':
                mfm(item[1] + '.conv_a', item[2], item[2], 1)
                mfm(item[1] + '.conv', item[2], item[3], 3)
        specs.append(('fc.weight', (256, 8 * 8 * 128), 'fc_w'))
        specs.append(('fc.bias', (256,), 'fc_b'))
        if self.fc2_hooked:
            specs.append(('fc2.weight', (self.num_classes, 256), 'fc_w'))   # bias=False: lightcnn.py:229
        return specs

    def build_program(self):
        p = Program(self.in_shape)

        def mfm(t, pre, cout, k, pad):
            t = p.conv(t, pre + '.filter', 2 * cout, k, stride=1, pad=pad)   # lightcnn.py:53,59
            t = p.split(t)                                                    # lightcnn.py:61
            return p.g_maxhalves(t)                                           # lightcnn.py:62
        t = 0
        for item in self._mfms():
            if item[0] == 'mfm':
                t = mfm(t, item[1], item[3], item[4], item[5])
            elif item[0] == 'pool':
                t = p.g_add(p.maxpool(t, 2, 2), p.avgpool(t, 2, 2))           # lightcnn.py:252
            elif item[0] == 'res':
                res = t
                o = mfm(t, item[1] + '.conv1', item[2], 3, 1)
                o = mfm(o, item[1] + '.conv2', item[2], 3, 1)
                t = p.add(o, res)                                             # lightcnn.py:88
            elif item[0] == 'group':
                t = mfm(t, item[1] + '.conv_a', item[2], 1, 0)
                t = mfm(t, item[1] + '.conv', item[3], 3, 1)
        t = p.linear(t, 'fc', 256, (8, 8))
        p.mark('encode', t)                                                   # features: whitebox.py:128-129
        if self.fc2_hooked:
            t = p.linear(t, 'fc2', self.num_classes, (1, 1), bias=False)
            p.mark('classify', t)
        return p


def LightCNN_29Layers_v2(**kwargs):
    """lightcnn.py:295-298"""
    model = network_29layers_v2((1, 2, 3, 4), **kwargs)
    model.training = False
    return model


def Load_Checkpoint(weights_path):
    """lightcnn.py:300-303"""
    checkpoint = torch.load(weights_path, map_location='cpu')
    return {k[7:]: v for k, v in checkpoint['state_dict'].items()}
