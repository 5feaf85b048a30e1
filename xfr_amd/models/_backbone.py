"""Parameter container shared by the three backbones.

The reference backbones are nn.Modules whose forward is executed by PyTorch; here a backbone is only
(a) a named parameter store with the reference's state_dict keys, so its checkpoints load unchanged, and
(b) a builder for the static layer program the HIP engine executes.  There is deliberately no torch forward:
the product path is the engine, and it fails loudly if the engine cannot be loaded.
"""
from collections import OrderedDict

import torch


class Backbone(object):
    arch = None
    in_shape = None        # (C, H, W)

    def __init__(self):
        self._state = OrderedDict()
        self._device = None
        self.version = 0   # bumped whenever parameters change -> engines re-upload
        self.training = False

    # -- to be provided by subclasses -----------------------------------------------------------------
    def param_specs(self):
        """[(state_dict key, shape, kind)], kind in conv_w conv_b bn_w bn_b bn_mean bn_var bn_nbt fc_w fc_b."""
        raise NotImplementedError

    def build_program(self):
        raise NotImplementedError

    # -- nn.Module-like surface used by the reference's callers ---------------------------------------
    def init_parameters(self, generator=None):
        """Reference initialisers: resnet.py:191-198 for convs / BN; torch defaults are replaced by the
        same normal recipe for the other layers (only used when no checkpoint is given)."""
        import math
        g = generator
        for name, shape, kind in self.param_specs():
            if kind == 'bn_nbt':
                t = torch.zeros((), dtype=torch.int64)
            elif kind in ('bn_w', 'bn_var'):
                t = torch.ones(shape)
            elif kind in ('bn_b', 'bn_mean'):
                t = torch.zeros(shape)
            elif kind == 'conv_w':
                n = shape[2] * shape[3] * shape[0]
                t = torch.randn(shape, generator=g) * math.sqrt(2.0 / n)
            elif kind == 'conv_b':
                t = torch.zeros(shape)
            elif kind == 'fc_w':
                t = torch.randn(shape, generator=g) * math.sqrt(1.0 / shape[1])
            else:
                t = torch.zeros(shape)
            self._state[name] = t
        self.version += 1
        return self

    def state_dict(self):
        return OrderedDict(self._state)

    def load_state_dict(self, sd, strict=True):
        specs = self.param_specs()
        missing = [n for n, _, k in specs if n not in sd and k != 'bn_nbt']
        unexpected = [k for k in sd if k not in {n for n, _, _ in specs}]
        if strict and (missing or unexpected):
            raise RuntimeError('Error(s) in loading state_dict: missing %s unexpected %s' % (missing, unexpected))
        for name, shape, kind in specs:
            if name in sd:
                t = sd[name].detach().to('cpu')
                if kind != 'bn_nbt':
                    t = t.float().contiguous()
                    if tuple(t.shape) != tuple(shape):
                        raise RuntimeError('size mismatch for %s: %s vs %s' % (name, tuple(t.shape), tuple(shape)))
                self._state[name] = t.clone()
        self.version += 1
        return self

    def set_parameter(self, name, tensor):
        self._state[name] = tensor.detach().to('cpu').float().contiguous().clone()
        self.version += 1

    def remove_parameter(self, name):
        if name in self._state:
            del self._state[name]
            self.version += 1

    def parameters(self):
        for k, v in self._state.items():
            if v.dtype.is_floating_point:
                yield _DeviceTagged(v, self.device)

    @property
    def device(self):
        if self._device is not None:
            return self._device
        return torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')

    def to(self, device):
        self._device = torch.device(device) if device is not None else None
        return self

    def cuda(self, device=None):
        return self.to(torch.device('cuda', 0 if device is None else device))

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        self.training = bool(mode)
        return self


class _DeviceTagged(object):
    """What `next(net.parameters())` returns: host master copy + the device the engine runs on
    (the reference only ever reads `.device` / `.is_cuda` from it: whitebox.py:227,774)."""

    def __init__(self, t, device):
        self.data = t
        self.device = device
        self.is_cuda = (device.type == 'cuda')
        self.grad = None
        self.shape = t.shape
