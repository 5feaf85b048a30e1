"""Host logic: layer programs, parameter specs, preprocessing -- pinned by structure captured from the reference."""
import collections

import numpy as np
import PIL.Image
import pytest
import torch

import golden_cases as GC
from parity_utils import make_backbone
from xfr_amd.program import OpKind


@pytest.mark.parametrize('arch,nc', [('stresnet101', 65359), ('resnet50_128', None), ('lightcnn29v2', 80013)])
def test_state_dict_keys_and_shapes_match_reference(arch, nc):
    g = GC.golden('golden_structure')
    bb, sd = make_backbone(arch, num_classes=nc)
    ours = [(k, ','.join(str(s) for s in v.shape)) for k, v in sd.items()]
    ref = list(zip([str(k) for k in g['%s/keys' % arch]], [str(s) for s in g['%s/shapes' % arch]]))
    assert sorted(ours) == sorted(ref)


def firing_kinds(prog, upto):
    """Module class name per hook firing, reference order: descending producer, registration order, image last."""
    names = {OpKind.CONV: 'Conv2d', OpKind.BATCHNORM: 'BatchNorm2d', OpKind.RELU: 'ReLU', OpKind.MAXPOOL: 'MaxPool2d',
             OpKind.AVGPOOL: 'AvgPool2d', OpKind.ADD: 'Add', OpKind.CONCAT: 'ConcatChannels', OpKind.MULTIPLY: 'Multiply',
             OpKind.LINEAR: 'Linear', OpKind.SPLIT: 'Split'}
    hooks = collections.defaultdict(list)
    for o in prog.ops:
        if o.out > upto or o.kind >= OpKind.G_ADD:
            continue
        ins = [o.in0] + ([o.in1] if o.kind == OpKind.ADD else [])
        for t in ins:
            hooks[o.out if (o.kind == OpKind.RELU and o.inplace) else t].append(names[OpKind(o.kind)])
    out = []
    for t in range(upto, -1, -1):
        out += hooks.get(t, [])
    return out


def test_firing_order_matches_reference_traces():
    g = GC.golden('golden_r101')
    bb, _ = make_backbone('stresnet101', num_classes=65359)
    prog = bb.build_program()
    assert firing_kinds(prog, prog.marks['classify']) == [str(n) for n in g['r101/affineonly_with_prior/hooked/ebp/names']]
    assert firing_kinds(prog, prog.marks['encode']) == [str(n) for n in g['r101/affineonly_with_prior/triplet/ebp/names']]
    g = GC.golden('golden_r50')
    bb, _ = make_backbone('resnet50_128')
    prog = bb.build_program()
    assert firing_kinds(prog, prog.marks['encode']) == [str(n) for n in g['r50/norelu/triplet/ebp/names']]
    g = GC.golden('golden_lcnn')
    bb, _ = make_backbone('lightcnn29v2', num_classes=80013)
    prog = bb.build_program()
    assert firing_kinds(prog, prog.marks['classify']) == [str(n) for n in g['lcnn/affineonly/hooked/ebp/names']]


def test_hooked_call_counts():
    bb, _ = make_backbone('stresnet101', num_classes=65359)
    c = collections.Counter(OpKind(o.kind) for o in bb.build_program().ops)
    assert (c[OpKind.CONV], c[OpKind.BATCHNORM], c[OpKind.RELU], c[OpKind.ADD], c[OpKind.AVGPOOL], c[OpKind.CONCAT],
            c[OpKind.MAXPOOL], c[OpKind.MULTIPLY], c[OpKind.LINEAR]) == (100, 100, 100, 33, 5, 4, 1, 1, 2)
    bb, _ = make_backbone('resnet50_128')
    c = collections.Counter(OpKind(o.kind) for o in bb.build_program().ops)
    assert (c[OpKind.CONV], c[OpKind.BATCHNORM], c[OpKind.RELU], c[OpKind.MAXPOOL], c[OpKind.AVGPOOL], c[OpKind.G_ADD]) == (54, 53, 49, 1, 1, 16)
    bb, _ = make_backbone('lightcnn29v2', num_classes=80013)
    c = collections.Counter(OpKind(o.kind) for o in bb.build_program().ops)
    assert (c[OpKind.CONV], c[OpKind.SPLIT], c[OpKind.ADD], c[OpKind.MAXPOOL], c[OpKind.AVGPOOL], c[OpKind.LINEAR]) == (29, 29, 10, 4, 4, 2)


def test_preprocess_shapes_and_values():
    from xfr_amd.models import resnet, whitebox as WB
    from xfr_amd.models.lightcnn import lightcnn_preprocess
    rng = np.random.RandomState(0)
    im = PIL.Image.fromarray(rng.randint(0, 256, (300, 260, 3)).astype(np.uint8))
    bb, _ = make_backbone('stresnet_mini')
    x = WB.WhiteboxSTResnet(bb).preprocess(im)                      # whitebox.py:108-110
    assert tuple(x.shape) == (1, 3, 224, 224) and x.dtype == torch.float32
    want = np.moveaxis(np.array(im.resize((224, 224)).convert('RGB')) - resnet.MEAN_RGB, 2, 0)
    assert np.allclose(x[0].numpy(), want.astype(np.float32))
    bb50, _ = make_backbone('resnet50_128')
    x = WB.Whitebox_resnet50_128(bb50).preprocess(im)                # whitebox.py:235-258
    assert tuple(x.shape) == (1, 3, 224, 224)
    x = lightcnn_preprocess()(im)                                    # lightcnn.py:27-31
    assert tuple(x.shape) == (1, 1, 128, 128) and 0.0 <= float(x.min()) and float(x.max()) <= 1.0


def test_unsupported_layer_kind_is_rejected_by_name():
    """whitebox.py:402-403: Sigmoid / ELU / Tanh raise ValueError; the engine refuses unknown op kinds the same way."""
    import ctypes
    from xfr_amd import _lib
    from xfr_amd.program import Program
    p = Program((3, 32, 32))
    t = p.conv(0, 'c', 8, 3, 1, 1)
    p._op(99, t)     # not a kind the engine knows (e.g. Sigmoid)
    h = ctypes.c_void_p()
    lib = _lib.load()
    st = lib.xfr_engine_create(p.op_array(), len(p.ops), len(p.weight_names), 3, 32, 32, 1, 0, ctypes.byref(h))
    if torch.cuda.is_available():
        assert st == _lib.XFR_UNSUPPORTED_LAYER
        with pytest.raises(ValueError):
            _lib.check(st)
    else:
        assert st == _lib.XFR_HIP_ERROR


def test_create_wbnet_defaults_match_reference():
    """eval/create_wbnet.py:24-132: default subtree modes, thresholds, Platt scalings; unknown names raise."""
    import warnings
    from xfr_amd.create_wbnet import create_wbnet
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        wb = create_wbnet('resnetv4_pytorch', device='cuda:0')
        assert wb.ebp_subtree_mode() == 'norelu' and wb.match_threshold == 0.9722 and wb.platts_scaling == 16.61
        wb = create_wbnet('resnetv6_pytorch', device='cuda:0', ebp_subtree_mode='all')
        assert wb.ebp_subtree_mode() == 'all' and wb.match_threshold == 0.9636
        wb = create_wbnet('vggface2_resnet50', device='cuda:0')
        assert wb.ebp_subtree_mode() == 'norelu' and abs(wb.platts_scaling - 15.921608) < 1e-9
        wb = create_wbnet('lightcnn', device='cuda:0')
        assert wb.ebp_subtree_mode() == 'affineonly_with_prior' and wb.net.num_classes() == 80013
    with pytest.raises(NotImplementedError):
        create_wbnet('senet50', device='cuda:0')
    with pytest.raises(DeprecationWarning):
        create_wbnet('resnetv4_pytorch', device='cuda:0', ebp_version=3)


def test_layernames_equal_the_reference_strings():
    """Whitebox.P_layername (whitebox.py:393: str(module) of the hooked module of every firing, the image hook last): the lists the layer
    programs produce -- planner firing order (device-free xfr_plan_describe) + torch module reprs -- against the lists the REAL reference
    produced under this image's torch (tests/golden/make_golden_names.py), all three backbones, hooked and triplet classifier."""
    import numpy as np
    import torch
    from parity_utils import make_backbone
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_layernames.npz"))
    if str(gold['torch_version']).split('+')[0] != torch.__version__.split('+')[0]:
        pytest.skip('fixture was generated under torch %s' % gold['torch_version'])
    seen = 0
    for arch, ncls in (('stresnet_mini', 5), ('stresnet101', 7), ('resnet50_128', None), ('lightcnn29v2', 7)):
        bb, _ = make_backbone(arch, seed=1, num_classes=ncls)
        prog = bb.build_program()
        for cls, mark in (('hooked', 'classify'), ('triplet', 'encode')):
            key = arch + '/' + cls
            if key not in gold.files:
                continue
            for mode in ('affineonly_with_prior', 'norelu'):      # the firing order does not depend on the subtree mode
                assert prog.layernames(mode, prog.marks[mark]) == [str(x) for x in gold[key]], (key, mode)
            seen += 1
    assert seen == 7
