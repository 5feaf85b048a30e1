"""Static layer program of a backbone: the host-side description handed to the HIP engine.

The reference discovers the network dynamically (forward / pre-forward hooks on every leaf nn.Module,
whitebox.py:34-56,303) and lets autograd record the graph on every call.  Here the same information --
the sequence of hooked leaf-module calls and of the functional glue between them -- is written down once,
as an array of `xfr_op_desc` records (include/xfr_amd.h), in the reference's forward call order.
"""
import ctypes
from enum import IntEnum


class OpKind(IntEnum):
    CONV = 1
    BATCHNORM = 2
    RELU = 3
    MAXPOOL = 4
    AVGPOOL = 5
    ADD = 6
    CONCAT = 7
    MULTIPLY = 8
    LINEAR = 9
    SPLIT = 10
    G_ADD = 11
    G_MAXHALVES = 12
    G_NORMALIZE = 13


# str(module) class names of the reference modules, for Whitebox.P_layername (whitebox.py:393)
LAYER_NAMES = {
    OpKind.CONV: 'Conv2d', OpKind.BATCHNORM: 'BatchNorm2d', OpKind.RELU: 'ReLU', OpKind.MAXPOOL: 'MaxPool2d',
    OpKind.AVGPOOL: 'AvgPool2d', OpKind.ADD: 'Add', OpKind.CONCAT: 'ConcatChannels',
    OpKind.MULTIPLY: 'Multiply', OpKind.LINEAR: 'Linear', OpKind.SPLIT: 'Split',
}


class OpDesc(ctypes.Structure):
    """Mirror of `xfr_op_desc` in include/xfr_amd.h."""
    _fields_ = [
        ('kind', ctypes.c_int32), ('in0', ctypes.c_int32), ('in1', ctypes.c_int32), ('out', ctypes.c_int32),
        ('cout', ctypes.c_int32), ('kh', ctypes.c_int32), ('kw', ctypes.c_int32), ('stride', ctypes.c_int32),
        ('pad', ctypes.c_int32), ('ceil_mode', ctypes.c_int32), ('inplace', ctypes.c_int32),
        ('fparam', ctypes.c_float),
        ('w_weight', ctypes.c_int32), ('w_bias', ctypes.c_int32), ('w_mean', ctypes.c_int32),
        ('w_var', ctypes.c_int32),
    ]


class Program(object):
    """Builder for a layer program.  Tensor id 0 is the image; every op returns its output tensor id."""

    def __init__(self, in_shape):
        self.in_shape = tuple(int(v) for v in in_shape)  # (C, H, W)
        self.ops = []
        self.weight_names = []     # weight table: state_dict keys in table order
        self._widx = {}
        self.ntensors = 1
        self.reprs = {}            # op index -> str(module) override (module_repr)
        self.marks = {}            # named tensors: 'encode', 'classify'

    def _w(self, name):
        if name is None:
            return -1
        if name not in self._widx:
            self._widx[name] = len(self.weight_names)
            self.weight_names.append(name)
        return self._widx[name]

    def _op(self, kind, in0, in1=-1, cout=0, kh=0, kw=0, stride=1, pad=0, ceil_mode=0, inplace=0, fparam=0.0,
            w_weight=None, w_bias=None, w_mean=None, w_var=None):
        out = self.ntensors
        self.ntensors += 1
        self.ops.append(OpDesc(int(kind), int(in0), int(in1), out, int(cout), int(kh), int(kw), int(stride),
                               int(pad), int(ceil_mode), int(inplace), float(fparam),
                               self._w(w_weight), self._w(w_bias), self._w(w_mean), self._w(w_var)))
        return out

    # hooked leaf modules -----------------------------------------------------------------------------
    def conv(self, x, prefix, cout, k, stride=1, pad=0, bias=True):
        return self._op(OpKind.CONV, x, cout=cout, kh=k, kw=k, stride=stride, pad=pad,
                        w_weight=prefix + '.weight', w_bias=(prefix + '.bias') if bias else None)

    def batchnorm(self, x, prefix, eps=1e-5):
        return self._op(OpKind.BATCHNORM, x, fparam=eps, w_weight=prefix + '.weight', w_bias=prefix + '.bias',
                        w_mean=prefix + '.running_mean', w_var=prefix + '.running_var')

    def relu_(self, x):
        return self._op(OpKind.RELU, x, inplace=1)

    def maxpool(self, x, k, stride, pad=0, ceil_mode=False):
        return self._op(OpKind.MAXPOOL, x, kh=k, kw=k, stride=stride, pad=pad, ceil_mode=1 if ceil_mode else 0)

    def avgpool(self, x, k, stride):
        return self._op(OpKind.AVGPOOL, x, kh=k, kw=k, stride=stride)

    def add(self, a, b):
        return self._op(OpKind.ADD, a, b)

    def concat_channels(self, x, channels):
        return self._op(OpKind.CONCAT, x, cout=channels)

    def multiply(self, x, n):
        return self._op(OpKind.MULTIPLY, x, fparam=n)

    def linear(self, x, prefix, cout, in_hw, bias=True):
        """nn.Linear applied to the (C,H,W)-flattened tensor x: a convolution whose kernel is the whole
        spatial extent (PyTorch flattens in c,h,w order, which is the conv weight's own layout)."""
        return self._op(OpKind.LINEAR, x, cout=cout, kh=in_hw[0], kw=in_hw[1], stride=1, pad=0,
                        w_weight=prefix + '.weight', w_bias=(prefix + '.bias') if bias else None)

    def split(self, x):
        return self._op(OpKind.SPLIT, x)

    # glue ---------------------------------------------------------------------------------------------
    def g_add(self, a, b):
        return self._op(OpKind.G_ADD, a, b)

    def g_maxhalves(self, x):
        return self._op(OpKind.G_MAXHALVES, x)

    def g_normalize(self, x):
        return self._op(OpKind.G_NORMALIZE, x)

    def mark(self, name, tensor):
        self.marks[name] = tensor
        return tensor

    def op_array(self):
        arr = (OpDesc * len(self.ops))()
        for i, o in enumerate(self.ops):
            arr[i] = o
        return arr

    def describe(self, mode, seed_tensor, batch=32):
        """The planner's fused schedules for this program as text (xfr_plan_describe: no device needed)."""
        from . import _lib
        from .engine import MODES
        lib = _lib.load()
        c, h, w = self.in_shape
        need = ctypes.c_size_t()
        args = (self.op_array(), len(self.ops), len(self.weight_names), c, h, w, int(batch), MODES[mode], int(seed_tensor))
        _lib.check(lib.xfr_plan_describe(*args, None, 0, ctypes.byref(need)))
        buf = ctypes.create_string_buffer(need.value)
        _lib.check(lib.xfr_plan_describe(*args, buf, need.value, None))
        return buf.value.decode()

    def firing_ops(self, mode, seed_tensor):
        """Index (into self.ops) of the hooked module call behind every firing of a sweep seeded at `seed_tensor`, in the reference's firing
        order -- from the planner, no device needed."""
        for ln in self.describe(mode, seed_tensor, batch=1).splitlines():
            if ln.startswith('firing_ops'):
                return [int(v) for v in ln.split()[1:]]
        raise RuntimeError('xfr_plan_describe printed no firing_ops line')

    def tensor_channels(self):
        """Channel count of every tensor id (forward shape inference, channels only)."""
        ch = {0: self.in_shape[0]}
        for o in self.ops:
            k = OpKind(o.kind)
            if k in (OpKind.CONV, OpKind.LINEAR):
                ch[o.out] = o.cout
            elif k == OpKind.CONCAT:
                ch[o.out] = ch[o.in0] * (1 + o.cout)
            elif k == OpKind.G_MAXHALVES:
                ch[o.out] = ch[o.in0] // 2
            else:
                ch[o.out] = ch[o.in0]
        return ch

    def module_repr(self, k, ch=None):
        """str(module) of hooked op k as THIS torch prints it (Whitebox.P_layername, whitebox.py:393): the string is built by instantiating
        the same torch module on the meta device; the custom modules of the reference (Add, ConcatChannels, Multiply, Split: resnet.py:104-166,
        lightcnn.py:33-46) define no extra_repr.  A backbone whose reference module was built with other argument forms (resnet50_128's pools take
        lists) overrides the string in self.reprs[k]."""
        if k in self.reprs:
            return self.reprs[k]
        import torch
        o = self.ops[k]
        kind = OpKind(o.kind)
        ch = ch or self.tensor_channels()
        with torch.device('meta'):
            if kind == OpKind.CONV:
                return str(torch.nn.Conv2d(ch[o.in0], o.cout, o.kh, stride=o.stride, padding=o.pad, bias=o.w_bias >= 0))
            if kind == OpKind.BATCHNORM:
                return str(torch.nn.BatchNorm2d(ch[o.in0], eps=float('%.6g' % o.fparam)))
            if kind == OpKind.RELU:
                return str(torch.nn.ReLU(inplace=bool(o.inplace)))
            if kind == OpKind.MAXPOOL:
                return str(torch.nn.MaxPool2d(o.kh, o.stride, padding=o.pad, ceil_mode=bool(o.ceil_mode)))
            if kind == OpKind.AVGPOOL:
                return str(torch.nn.AvgPool2d(o.kh, o.stride))
            if kind == OpKind.LINEAR:
                return str(torch.nn.Linear(ch[o.in0] * o.kh * o.kw, o.cout, bias=o.w_bias >= 0))
        return LAYER_NAMES[kind] + '()'

    def layernames(self, mode, seed_tensor):
        """Whitebox.P_layername of a sweep (whitebox.py:393), image hook included (last entry, like the reference's list)."""
        ch = self.tensor_channels()
        return [self.module_repr(k, ch) for k in self.firing_ops(mode, seed_tensor)] + [self.module_repr(0, ch)]

    def hooked_kinds(self):
        return [OpKind(o.kind) for o in self.ops if o.kind < OpKind.G_ADD]
