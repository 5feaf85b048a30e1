"""The C-ABI boundary: the shared library loads and exports exactly what include/xfr_amd.h declares, the ctypes
mirror of the structs matches, and the product path fails loudly without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

from xfr_amd import _lib
from xfr_amd.program import OpDesc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'xfr_amd.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(xfr_[a-z_0-9]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = header_functions()
    bound = sorted(n for n, _, _ in _lib.SYMBOLS)
    assert declared == bound, 'header and ctypes binding disagree: %s' % (set(declared) ^ set(bound))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.xfr_abi_version() == _lib.ABI_VERSION


def test_opdesc_layout():
    assert ctypes.sizeof(OpDesc) == 16 * 4
    assert OpDesc.fparam.offset == 11 * 4 and OpDesc.w_var.offset == 15 * 4


def test_error_strings_and_arg_checks():
    lib = _lib.load()
    h = ctypes.c_void_p()
    st = lib.xfr_engine_create(None, 0, 0, 3, 224, 224, 1, 0, ctypes.byref(h))
    assert st == _lib.XFR_INVALID_ARG
    assert b'bad arguments' in lib.xfr_last_error()
    assert lib.xfr_engine_set_mode(None, 1, 1e-16, 0) == _lib.XFR_INVALID_ARG


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_no_gpu_means_loud_failure_not_fallback():
    from parity_utils import make_backbone, make_images
    from xfr_amd.models import whitebox as WB
    bb, _ = make_backbone('stresnet_mini')
    wb = WB.Whitebox(WB.WhiteboxSTResnet(bb))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        wb.net.encode(make_images('stresnet_mini', 1))
    prog = bb.build_program()
    h = ctypes.c_void_p()
    st = _lib.load().xfr_engine_create(prog.op_array(), len(prog.ops), len(prog.weight_names), 3, 224, 224, 1, 0, ctypes.byref(h))
    assert st == _lib.XFR_HIP_ERROR and b'no CPU fallback' in _lib.load().xfr_last_error()


def test_whitebox_constructor_errors_match_reference():
    from parity_utils import make_backbone
    from xfr_amd.models import whitebox as WB
    bb, _ = make_backbone('stresnet_mini')
    net = WB.WhiteboxSTResnet(bb)
    with pytest.raises(RuntimeError, match='at least 4'):        # whitebox.py:283-284
        WB.Whitebox(net, ebp_version=3)
    with pytest.raises(ValueError, match='Invalid subtree mode'):  # whitebox.py:430
        WB.Whitebox(net, ebp_subtree_mode='bogus')
    with pytest.raises(AssertionError):                           # whitebox.py:275
        WB.Whitebox(bb)
    wb = WB.Whitebox(net, ebp_version=11)
    assert wb._ebp_with_bias is True and wb.convert_saliency_uint8 is True      # whitebox.py:285-289
    assert WB.Whitebox(net).ebp_subtree_mode() == 'affineonly_with_prior' and WB.Whitebox(net).eps == 1e-16


def _build_c_host(tmp_path):
    import subprocess
    exe = str(tmp_path / 'c_host')
    csrc = os.path.join(ROOT, 'xfr_amd', 'csrc')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Wextra', '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', 'c_host.c'),
                           '-L' + csrc, '-lxfr_amd', '-Wl,-rpath,' + csrc, '-ldl', '-lm', '-o', exe])
    return exe


def test_header_is_plain_c_and_a_c_host_links(tmp_path):
    """include/xfr_amd.h is C99 (no C++ or torch types at the boundary): examples/c_host.c compiles with gcc -std=c99 against
    it, links against libxfr_amd.so alone and runs -- the planner everywhere, the engine where a HIP device is visible."""
    import subprocess
    exe = _build_c_host(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'fused schedule' in out.stdout and 'bwd CONV_BWD' in out.stdout
    if not torch.cuda.is_available():
        assert 'no CPU fallback' in out.stdout


@pytest.mark.gpu
def test_c_host_runs_contrastive_ebp(tmp_path):
    import re
    import subprocess
    out = subprocess.run([_build_c_host(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r'saliency map 16x16: sum ([0-9.]+), max ([0-9.]+)', out.stdout)
    assert m and abs(float(m.group(1)) - 1.0) < 1e-4 and float(m.group(2)) > 0
