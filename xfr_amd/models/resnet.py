"""STR-Janus ResNet-101: host-side mirror of python/xfr/models/resnet.py for the EBP hot path.

Same constructor names, state_dict keys and preprocessing as the reference; the arithmetic is the
layer program below, executed by the HIP engine (xfr_amd/csrc).
"""
import warnings

import numpy as np
import torch

from ..program import Program
from ._backbone import Backbone

MEAN_RGB = np.array(([122.782, 117.001, 104.298]))   # resnet.py:23


def convert_resnet101v4_image(img, mean_rgb=MEAN_RGB):
    """RGB byte image (PIL or HxWx3 ndarray) -> 3xHxW float tensor, mean subtracted (resnet.py:25-37)."""
    if isinstance(img, np.ndarray):
        img_fp = img - MEAN_RGB
    else:
        img_fp = img.convert('RGB') - MEAN_RGB
    img_fp = np.moveaxis(img_fp, 2, 0)
    return torch.from_numpy(img_fp).float()


def pil_to_tensor(img):
    return convert_resnet101v4_image(img)


class ResNet(Backbone):
    """resnet.py:168-265 with block = Bottleneck (resnet.py:111-149, expansion 4)."""
    arch = 'stresnet'
    in_shape = (3, 224, 224)
    expansion = 4

    def __init__(self, layers, mode='encode', num_classes=65359):
        super(ResNet, self).__init__()
        if mode not in {'encode', 'classify', 'both'}:
            raise Exception('mode should be one of ' + str({'encode', 'classify', 'both'}))
        self.mode = mode
        self.layers = tuple(layers)
        self.num_classes = int(num_classes)
        self.fc2_hooked = True     # False after WhiteboxSTResnet.set_triplet_classifier (whitebox.py:93-96)
        self.init_parameters()

    def _blocks(self):
        inplanes = 64
        for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), self.layers)):
            stride0 = 1 if li == 0 else 2
            for b in range(blocks):
                stride = stride0 if b == 0 else 1
                downsample = (b == 0) and (stride != 1 or inplanes != planes * self.expansion)  # resnet.py:202
                yield 'layer%d.%d.' % (li + 1, b), inplanes, planes, stride, downsample
                inplanes = planes * self.expansion

    def param_specs(self):
        specs = []

        def conv(p, cin, cout, k):
            specs.append((p + '.weight', (cout, cin, k, k), 'conv_w'))
            specs.append((p + '.bias', (cout,), 'conv_b'))     # bias=True everywhere: resnet.py:116-122,177

        def bn(p, c):
            specs.extend([(p + '.weight', (c,), 'bn_w'), (p + '.bias', (c,), 'bn_b'),
                          (p + '.running_mean', (c,), 'bn_mean'), (p + '.running_var', (c,), 'bn_var'),
                          (p + '.num_batches_tracked', (), 'bn_nbt')])
        conv('conv1', 3, 64, 7)
        bn('bn1', 64)
        for pre, inplanes, planes, stride, downsample in self._blocks():
            conv(pre + 'conv1', inplanes, planes, 1)
            bn(pre + 'bn1', planes)
            conv(pre + 'conv2', planes, planes, 3)
            bn(pre + 'bn2', planes)
            conv(pre + 'conv3', planes, planes * 4, 1)
            bn(pre + 'bn3', planes * 4)
        specs.append(('fc1.weight', (512, 512 * self.expansion), 'fc_w'))
        specs.append(('fc1.bias', (512,), 'fc_b'))
        if self.fc2_hooked:
            specs.append(('fc2.weight', (self.num_classes, 512), 'fc_w'))
            specs.append(('fc2.bias', (self.num_classes,), 'fc_b'))
        return specs

    def build_program(self):
        """ResNet.forward resnet.py:224-265; Bottleneck.forward :129-149; downsample :210-213."""
        p = Program(self.in_shape)
        t = p.conv(0, 'conv1', 64, 7, stride=2, pad=3)
        t = p.batchnorm(t, 'bn1')
        t = p.relu_(t)
        t = p.maxpool(t, 3, 2, 1)
        for pre, inplanes, planes, stride, downsample in self._blocks():
            residual = t
            o = p.conv(t, pre + 'conv1', planes, 1, stride=stride)
            o = p.batchnorm(o, pre + 'bn1')
            o = p.relu_(o)
            o = p.conv(o, pre + 'conv2', planes, 3, stride=1, pad=1)
            o = p.batchnorm(o, pre + 'bn2')
            o = p.relu_(o)
            o = p.conv(o, pre + 'conv3', planes * 4, 1)
            o = p.batchnorm(o, pre + 'bn3')
            if downsample:
                residual = p.avgpool(t, stride, stride)
                residual = p.concat_channels(residual, planes * 4 // inplanes - 1)
            t = p.relu_(p.add(o, residual))
        t = p.avgpool(t, 7, 7)
        t = p.linear(t, 'fc1', 512, (1, 1))
        t = p.g_normalize(t)
        t = p.multiply(t, 50.0)
        p.mark('encode', t)
        if self.fc2_hooked:
            t = p.linear(t, 'fc2', self.num_classes, (1, 1))
            p.mark('classify', t)
        return p


def resnet101v6(pthfile, device=None):
    """resnet.py:268-279.  pthfile=None keeps the constructor's random initialisation (no checkpoint ships
    with the reference: models/*.pth are git-LFS pointers)."""
    if device is None and not torch.cuda.is_available():
        warnings.warn('no HIP device visible: the xfr_amd engine cannot run in this process')
    model = ResNet([3, 4, 23, 3], mode='encode', num_classes=65359)
    if pthfile is not None:
        model.load_state_dict(torch.load(pthfile, map_location='cpu'))
    if device is not None:
        model.to(device)
    return model


def stresnet101(pthfile, device=None):
    return resnet101v6(pthfile, device)
