"""Host logic of the engine that needs no GPU: the planner (xfr_plan_describe) on the three backbones, and the compiled-epilogue
signature table (xfr_amd/csrc/chain_sigs.inc) against it."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from xfr_amd import _lib
from xfr_amd.models import lightcnn, resnet, resnet50_128

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = ('affineonly', 'affineonly_with_prior', 'norelu', 'all')



def sig_code(op, s0=7, s1=7, store=0, step=0):
    """One step of a chain signature (xfr_amd/csrc/common.h): op | s0 << 5 | s1 << 8 | store << 11 | step << 12."""
    return op | (s0 << 5) | (s1 << 8) | (store << 11) | (step << 12)


def _programs():
    return {'stresnet101': resnet.ResNet([3, 4, 23, 3], num_classes=65359).build_program(),
            'stresnet_mini': resnet.ResNet([1, 1, 1, 1], num_classes=5).build_program(),
            'resnet50_128': resnet50_128.Resnet50_128().build_program(),
            'lightcnn29v2': lightcnn.LightCNN_29Layers_v2(num_classes=80013).build_program()}


PROGRAMS = _programs()


@pytest.mark.parametrize('arch', sorted(PROGRAMS))
def test_every_fused_chain_has_a_compiled_epilogue(arch):
    """Every elementwise chain the planner fuses behind a GEMM -- all four subtree modes, hooked and triplet seed -- is in the
    committed signature table, i.e. no launch of the three BASELINE backbones falls back to the interpreted epilogue."""
    prog = PROGRAMS[arch]
    n_sig = 0
    for mode in MODES:
        for mark in ('encode', 'classify'):
            if mark not in prog.marks:
                continue
            text = prog.describe(mode, prog.marks[mark])
            sig_lines = [ln for ln in text.splitlines() if ' SIG ' in ln]
            missing = [ln for ln in sig_lines if ln.endswith('compiled=-1')]
            assert not missing, '%s/%s/%s: run tools/gen_chain_sigs.py and rebuild\n%s' % (arch, mode, mark, '\n'.join(missing[:5]))
            n_sig += len(sig_lines)
    assert n_sig > 0


def test_signature_table_is_what_the_generator_produces(tmp_path):
    """chain_sigs.inc is generated: regenerating it from the current planner gives the committed file."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import gen_chain_sigs as G
    committed = open(G.OUT).read()
    old = G.OUT
    try:
        G.OUT = str(tmp_path / 'chain_sigs.inc')
        G.main()
        assert open(G.OUT).read() == committed
    finally:
        G.OUT = old


def test_plan_shapes_resnet101():
    """Firing count and schedule size of the ResNet-101 triplet sweep (378 entries of Whitebox.P, the image hook
    is not computed); fusion brings the 321 backward launches of the literal schedule down to 135."""
    prog = PROGRAMS['stresnet101']
    text = prog.describe('affineonly_with_prior', prog.marks['encode'])
    m = re.match(r'plan seed_tensor (\d+) mode 1 firings (\d+) launches (\d+) \(unfused (\d+)\)', text)
    assert m and int(m.group(2)) == 377
    assert int(m.group(3)) < int(m.group(4)) // 2
    assert sum(ln.startswith('fwd CONV') for ln in text.splitlines()) == 100          # every convolution takes its BatchNorm (the fc has none)
    assert sum(ln.startswith('bwd CONV_BWD') for ln in text.splitlines()) == 100       # 100 convs + fc, minus the first layer's backward-data GEMM


def test_residual_blocks_leave_no_glue_launches():
    """The Add VJP's gradient copy forwards into the in-place hook flush behind it, and a backward GEMM merges with a chain that
    stores its intermediate value back (fuse_plan): a Light-CNN residual block is two GEMM launches, and the ResNet-101 sweep
    keeps no copy in front of a hook flush."""
    prog = PROGRAMS['lightcnn29v2']
    lines = prog.describe('affineonly', prog.marks['encode']).splitlines()
    bwd = [ln.split()[1] for ln in lines if ln.startswith('bwd ')]
    # the twelve residual blocks' sweeps: GEMM, GEMM, GEMM, ... with nothing in between except around the four pooling stages
    runs, cur = [], 0
    for k in bwd:
        if k == 'CONV_BWD':
            cur += 1
        else:
            runs.append(cur)
            cur = 0
    runs.append(cur)
    assert max(runs) >= 8, runs
    assert 'COPY' not in bwd, bwd
    assert any(' %04x' % sig_code(13, step=5) in ln for ln in lines)          # store-back + fan-out chains behind a GEMM exist and ...
    assert all('compiled=-1' not in ln for ln in lines)    # ... every one of them has a compiled epilogue
    prog = PROGRAMS['stresnet101']
    lines = prog.describe('affineonly_with_prior', prog.marks['encode']).splitlines()
    kinds = [ln.split()[1] for ln in lines if ln.startswith('bwd ')]
    # the slice copy of the stride-1 first block (layer 1) is forwarded into its hook launch; the three down-sampling blocks' copies, pooled hooks, average-pool VJPs and
    # scattering GEMMs are the head of the block-input chain (EW_AVGUP_IN), and no stand-alone chain follows a first block's Add-output GEMM
    assert kinds.count('COPY') == 0, kinds
    assert kinds.count('AVGPOOL_BWD') == 1, kinds        # the global pool in front of the embedding
    assert kinds.count('EW') == 7, kinds             # seed, top, three block inputs, the layer-1 slice hook (reads the slice in place), the stem
    m = re.match(r'plan seed_tensor (\d+) mode 1 firings (\d+) launches (\d+)', lines[0])
    assert int(m.group(3)) <= 110
    # with the switch off (epilogue-fusion bit 6) the separate launches are back, every GEMM chain still compiled
    os.environ['XFR_DESCRIBE_FUSION'] = '67'
    try:
        lines67 = prog.describe('affineonly_with_prior', prog.marks['encode']).splitlines()
    finally:
        del os.environ['XFR_DESCRIBE_FUSION']
    kinds67 = [ln.split()[1] for ln in lines67 if ln.startswith('bwd ')]
    assert kinds67.count('COPY') == 4 and kinds67.count('AVGPOOL_BWD') == 4, kinds67
    assert all('compiled=-1' not in ln for ln in lines67)


def test_downsampling_blocks_take_their_add_into_the_convolution_epilogue():
    """The reference computes a down-sampling block's shortcut behind the main path (resnet.py:144-146); the engine runs it first, so that all 33
    residual adds of the STR-ResNet-101 -- not only the 29 of the ordinary blocks -- sit in the epilogue of the block's last convolution, in the
    forward-only schedule and in the probe forward.  Epilogue-fusion bit 8 keeps program order and the four separate adds."""
    prog = PROGRAMS['stresnet101']

    def adds(lines, kind):
        return sum(ln.startswith(kind + ' CONV') and ' %04x' % sig_code(9, s0=3, step=1) in ln.split('SIG')[1] for ln in lines)     # EW_ADDP (prefetch slot 3) as step 1 / 2
    def adds_probe(lines):
        return sum(ln.startswith('probe CONV') and ' %04x' % sig_code(9, s0=3, step=2) in ln for ln in lines)
    lines = prog.describe('affineonly_with_prior', prog.marks['encode']).splitlines()
    assert adds(lines, 'fwd') == 33 and adds_probe(lines) == 33
    os.environ['XFR_DESCRIBE_FUSION'] = '259'
    try:
        lines259 = prog.describe('affineonly_with_prior', prog.marks['encode']).splitlines()
    finally:
        del os.environ['XFR_DESCRIBE_FUSION']
    assert adds(lines259, 'fwd') == 29 and adds_probe(lines259) == 29
    assert all('compiled=-1' not in ln for ln in lines + lines259)


def test_projection_shortcut_blocks_branch_in_the_gemm_epilogue():
    """ResNet-50-128d (projection shortcuts): the main path's hook chain of a stage's first block runs as a side branch of the Add-output GEMM's
    epilogue (EW_STORE actions 1 / 2, fuse_plan 3c): 60 backward launches, no stand-alone two-step chain, every chain compiled; with the switch
    off (epilogue-fusion bit 7) the four launches are back, still compiled."""
    prog = PROGRAMS['resnet50_128']
    for mode in ('norelu', 'affineonly', 'affineonly_with_prior', 'all'):
        lines = prog.describe(mode, prog.marks['encode'], batch=64).splitlines()
        m = re.match(r'plan seed_tensor (\d+) mode (\d) firings (\d+) launches (\d+)', lines[0])
        assert int(m.group(4)) == 60, lines[0]
        bwd = [ln for ln in lines if ln.startswith('bwd ')]
        assert not [ln for ln in bwd if ln.split()[1] == 'EW' and ln.rstrip().endswith('steps 2')], mode
        assert all('compiled=-1' not in ln for ln in lines), mode
    os.environ['XFR_DESCRIBE_FUSION'] = '131'
    try:
        lines131 = prog.describe('norelu', prog.marks['encode'], batch=64).splitlines()
    finally:
        del os.environ['XFR_DESCRIBE_FUSION']
    assert ' launches 64 ' in lines131[0] and all('compiled=-1' not in ln for ln in lines131)
    assert len([ln for ln in lines131 if ln.startswith('bwd EW') and ln.rstrip().endswith('steps 2')]) == 4


def test_plan_describe_argument_checks():
    lib = _lib.load()
    prog = PROGRAMS['stresnet_mini']
    need = ctypes.c_size_t()
    ops = prog.op_array()
    assert lib.xfr_plan_describe(None, 0, 0, 3, 224, 224, 1, 1, 5, None, 0, ctypes.byref(need)) == _lib.XFR_INVALID_ARG
    assert lib.xfr_plan_describe(ops, len(prog.ops), len(prog.weight_names), 3, 224, 224, 1, 9, 5, None, 0, ctypes.byref(need)) == _lib.XFR_INVALID_ARG
    assert b'Invalid subtree mode' in lib.xfr_last_error()
    assert lib.xfr_plan_describe(ops, len(prog.ops), len(prog.weight_names), 3, 224, 224, 1, 1, 10 ** 6, None, 0, ctypes.byref(need)) == _lib.XFR_INVALID_ARG
    st = lib.xfr_plan_describe(ops, len(prog.ops), len(prog.weight_names), 3, 224, 224, 4, 1, prog.marks['encode'], None, 0, ctypes.byref(need))
    assert st == _lib.XFR_OK and need.value > 100
    buf = ctypes.create_string_buffer(32)          # truncation is safe and terminated
    assert lib.xfr_plan_describe(ops, len(prog.ops), len(prog.weight_names), 3, 224, 224, 4, 1, prog.marks['encode'], buf, 32, None) == _lib.XFR_OK
    assert len(buf.value) == 31


def test_tuning_hook_argument_checks():
    """xfr_debug_conv_stamps: a buffer needs a capacity -- positive (one record per workgroup index) or <= -256 (sampled mode, regions
    of 256 records per launch); NULL switches the hook off whatever the capacity says.  No device is touched."""
    lib = _lib.load()
    fake = ctypes.c_void_p(0x1000)
    assert lib.xfr_debug_conv_stamps(fake, 0) == _lib.XFR_INVALID_ARG
    assert lib.xfr_debug_conv_stamps(fake, -255) == _lib.XFR_INVALID_ARG
    assert b'xfr_debug_conv_stamps' in lib.xfr_last_error()
    assert lib.xfr_debug_conv_stamps(fake, -512) == _lib.XFR_OK
    assert lib.xfr_debug_conv_stamps(fake, 16) == _lib.XFR_OK
    assert lib.xfr_debug_conv_stamps(None, 0) == _lib.XFR_OK          # off again: nothing was launched in between


def test_inplace_relu_with_second_reader_is_rejected():
    """An in-place ReLU may not share its input with another consumer, whichever comes first in call order."""
    from xfr_amd.program import Program
    p = Program((3, 8, 8))
    c = p.conv(0, 'c1', 4, 3, pad=1)
    r = p.relu_(c)
    p.add(r, c)                     # reads the pre-ReLU tensor AFTER the in-place ReLU overwrote it
    need = ctypes.c_size_t()
    st = _lib.load().xfr_plan_describe(p.op_array(), len(p.ops), len(p.weight_names), 3, 8, 8, 1, 1, 2, None, 0, ctypes.byref(need))
    assert st == _lib.XFR_UNSUPPORTED_LAYER and b'in-place ReLU' in _lib.load().xfr_last_error()


def test_comm_entry_points_check_arguments():
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.xfr_comm_init(0, 0, None, 0, ctypes.byref(h)) == _lib.XFR_INVALID_ARG
    assert lib.xfr_broadcast_weights(None, None, 0, None) == _lib.XFR_INVALID_ARG
    assert lib.xfr_comm_destroy(None) == _lib.XFR_OK


# (backbone, subtree mode, xfr_engine_set_epilogue_fusion level or None = default) -> firings, launches of the fused sweep, launches of the un-fused one,
# backward GEMMs, stand-alone chain launches, gradient copies left, lean convolutions.  One row per fusion pass that can be switched: 67 = without the
# average-pool shortcut rewrite (pass 2b), 131 = without the projection-shortcut side branch (3c), 259 = the shortcut in program order (forward side
# only: the sweep is the default's), 0 = no fusion into GEMM epilogues at all (copy forwarding, the pool pair and chain -> chain merges remain).
FUSION_TABLE = [
    ('stresnet101', 'affineonly_with_prior', None, 377, 110, 321, 100, 7, 0, 48),
    ('stresnet101', 'norelu', None, 377, 110, 321, 100, 7, 0, 48),
    ('resnet50_128', 'affineonly_with_prior', None, 157, 60, 161, 53, 5, 0, 34),
    ('resnet50_128', 'norelu', None, 157, 60, 161, 53, 5, 0, 34),
    ('lightcnn29v2', 'affineonly_with_prior', None, 86, 33, 166, 29, 4, 0, 0),
    ('lightcnn29v2', 'norelu', None, 86, 33, 166, 29, 4, 0, 0),
    ('stresnet101', 'affineonly_with_prior', 67, 377, 120, 321, 100, 10, 4, 48),
    ('resnet50_128', 'norelu', 67, 157, 63, 161, 53, 5, 0, 34),
    ('stresnet101', 'norelu', 131, 377, 114, 321, 100, 11, 0, 48),
    ('resnet50_128', 'norelu', 131, 157, 64, 161, 53, 9, 0, 34),
    ('stresnet101', 'affineonly_with_prior', 259, 377, 110, 321, 100, 7, 0, 48),
    ('stresnet101', 'affineonly_with_prior', 0, 377, 124, 321, 100, 14, 4, 0),
    ('resnet50_128', 'norelu', 0, 157, 67, 161, 53, 9, 0, 0),
    ('lightcnn29v2', 'affineonly_with_prior', 0, 86, 45, 166, 29, 8, 0, 0),
]


@pytest.mark.parametrize('row', FUSION_TABLE, ids=lambda r: '%s-%s-%s' % (r[0], r[1], r[2]))
def test_fusion_passes_table(row):
    """The planner's fusion passes (engine.hip: fuse_copy_forwarding, fuse_pool_pair, fuse_downsample_avgpool / _projection, fuse_stage_head_relu /
    _branch, fuse_merge_to_fixed_point; lean_prepare), one table row per backbone / mode / switchable pass: what each pass removes is pinned as a
    launch count, so a change to one pass shows up as the row it moves."""
    import re
    from xfr_amd.models import lightcnn, resnet, resnet50_128
    arch, mode, fusion, firings, launches, unfused, gemms, chains, copies, lean = row
    prog = {'stresnet101': lambda: resnet.ResNet([3, 4, 23, 3], num_classes=65359), 'resnet50_128': resnet50_128.Resnet50_128,
            'lightcnn29v2': lambda: lightcnn.LightCNN_29Layers_v2(num_classes=80013)}[arch]().build_program()
    if fusion is not None:
        os.environ['XFR_DESCRIBE_FUSION'] = str(fusion)
    try:
        text = prog.describe(mode, prog.marks['encode'])
    finally:
        os.environ.pop('XFR_DESCRIBE_FUSION', None)
    m = re.search(r'firings (\d+) launches (\d+) \(unfused (\d+)\)', text)
    kinds = {}
    for ln in text.splitlines():
        if ln.startswith('bwd '):
            kinds[ln.split()[1]] = kinds.get(ln.split()[1], 0) + 1
    ml = re.search(r'lean convolutions (\d+)', text)
    got = (int(m.group(1)), int(m.group(2)), int(m.group(3)), kinds.get('CONV_BWD', 0), kinds.get('EW', 0), kinds.get('COPY', 0), int(ml.group(1)) if ml else 0)
    assert got == (firings, launches, unfused, gemms, chains, copies, lean)

