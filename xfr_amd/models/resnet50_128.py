"""VGGFace2 ResNet-50-128d: mirror of models/resnet50_128_pytorch/resnet50_128.py for the EBP hot path."""
import torch

from ..program import Program
from ._backbone import Backbone

_STAGES = ((2, 3, 64, 256), (3, 4, 128, 512), (4, 6, 256, 1024), (5, 3, 512, 2048))


class Resnet50_128(Backbone):
    arch = 'resnet50_128'
    in_shape = (3, 224, 224)

    def __init__(self):
        super(Resnet50_128, self).__init__()
        self.meta = {'mean': [131.0912, 103.8827, 91.4953], 'std': [1, 1, 1], 'imageSize': [224, 224, 3]}
        self.init_parameters()

    def _blocks(self):
        cin = 64
        for (s, nblocks, mid, outc) in _STAGES:
            for b in range(1, nblocks + 1):
                stride = 2 if (b == 1 and s > 2) else 1     # resnet50_128.py:46,54,84,92,140,148
                yield 'conv%d_%d' % (s, b), cin, mid, outc, stride, (b == 1)
                cin = outc

    def param_specs(self):
        specs = []

        def conv(p, cin, cout, k):
            specs.append((p + '.weight', (cout, cin, k, k), 'conv_w'))   # bias=False: resnet50_128.py:13

        def bn(p, c):
            specs.extend([(p + '.weight', (c,), 'bn_w'), (p + '.bias', (c,), 'bn_b'),
                          (p + '.running_mean', (c,), 'bn_mean'), (p + '.running_var', (c,), 'bn_var'),
                          (p + '.num_batches_tracked', (), 'bn_nbt')])
        conv('conv1_7x7_s2', 3, 64, 7)
        bn('conv1_7x7_s2_bn', 64)
        for pre, cin, mid, outc, stride, first in self._blocks():
            conv(pre + '_1x1_reduce', cin, mid, 1)
            bn(pre + '_1x1_reduce_bn', mid)
            conv(pre + '_3x3', mid, mid, 3)
            bn(pre + '_3x3_bn', mid)
            conv(pre + '_1x1_increase', mid, outc, 1)
            bn(pre + '_1x1_increase_bn', outc)
            if first:
                conv(pre + '_1x1_proj', cin, outc, 1)
                bn(pre + '_1x1_proj_bn', outc)
        conv('feat_extract', 2048, 128, 1)
        return specs

    def build_program(self):
        """Resnet50_128.forward resnet50_128.py:172-348."""
        p = Program(self.in_shape)
        t = p.conv(0, 'conv1_7x7_s2', 64, 7, stride=2, pad=3, bias=False)
        t = p.batchnorm(t, 'conv1_7x7_s2_bn')
        t = p.relu_(t)
        t = p.maxpool(t, 3, 2, 0, ceil_mode=True)               # resnet50_128.py:16
        # the reference passes lists / tuples to the pools, and torch prints them as given (Whitebox.P_layername is str(module))
        p.reprs[len(p.ops) - 1] = 'MaxPool2d(kernel_size=[3, 3], stride=[2, 2], padding=(0, 0), dilation=1, ceil_mode=True)'
        for pre, cin, mid, outc, stride, first in self._blocks():
            block_in = t
            o = p.conv(t, pre + '_1x1_reduce', mid, 1, stride=stride, bias=False)
            o = p.batchnorm(o, pre + '_1x1_reduce_bn')
            o = p.relu_(o)
            o = p.conv(o, pre + '_3x3', mid, 3, stride=1, pad=1, bias=False)
            o = p.batchnorm(o, pre + '_3x3_bn')
            o = p.relu_(o)
            o = p.conv(o, pre + '_1x1_increase', outc, 1, bias=False)
            o = p.batchnorm(o, pre + '_1x1_increase_bn')
            if first:
                sc = p.conv(block_in, pre + '_1x1_proj', outc, 1, stride=stride, bias=False)
                sc = p.batchnorm(sc, pre + '_1x1_proj_bn')
            else:
                sc = block_in
            t = p.g_add(sc, o)                                   # functional torch.add: resnet50_128.py:187
            t = p.relu_(t)
        t = p.avgpool(t, 7, 1)
        p.reprs[len(p.ops) - 1] = 'AvgPool2d(kernel_size=[7, 7], stride=[1, 1], padding=0)'
        t = p.conv(t, 'feat_extract', 128, 1, bias=False)
        p.mark('encode', t)                                      # net(x)[0]: whitebox.py:224
        return p


def resnet50_128(weights_path=None, **kwargs):
    """resnet50_128.py:350-361"""
    model = Resnet50_128()
    if weights_path:
        model.load_state_dict(torch.load(weights_path, map_location='cpu'))
    return model
