"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI (ctypes -> libxfr_amd.so);
the HIP engine is compared with the golden vectors captured from the real reference and with the CPU oracle."""
import ctypes
import os

import numpy as np
import pytest
import torch

import golden_cases as GC
from parity_utils import (MAP_RTOL, MAP_RTOL_CONTRAST, assert_map_close, assert_map_close_robust, assert_trace_close, emb_dim, make_backbone,
                          make_images, map_metrics)
from xfr_amd import synth

pytestmark = pytest.mark.gpu


LEAN_RTOL_CONTRAST = 2e-3       # lean vs literal schedule on contrastive maps: measured 4e-5 .. 1.1e-3 over backbones / modes (one ulp per hook, amplified by the contrast)
PARITY_REPORT = {}          # key -> measured margins of every golden case this session (tests/conftest.py writes parity_report.json)


def _check_factory():
    def check(key, res, trace, gold):
        want = gold[key + '/map']
        # Every golden case of every backbone is held to the STRICT criterion (measured on MI355X in round 2: all pass); only the
        # final maps of *truncated* calls ride on the percentile mask, a discontinuous step (see parity_utils docstring).
        # Contrastive maps whose classifier rows are two face encodings (cosine 0.9998 under seeded weights) get MAP_RTOL_CONTRAST; the
        # WELL-CONDITIONED ones (`synthetic/...`: independent random rows) are held to SURVEY.md section 8c's 1e-3.
        well = '/synthetic/' in key
        rel, cos = map_metrics(res, want)
        rec = {'max_abs_diff_over_max': float(rel), 'cosine': float(cos)}
        if key.endswith('truncated'):
            rec['criterion'] = 'robust %.0e' % (MAP_RTOL if well else MAP_RTOL_CONTRAST)
            PARITY_REPORT[key] = rec
            assert_map_close_robust(res, want, key, rtol=MAP_RTOL if well else MAP_RTOL_CONTRAST)
        elif key.endswith('contrastive'):
            rec['criterion'] = 'strict %.0e' % (MAP_RTOL if well else MAP_RTOL_CONTRAST)
            PARITY_REPORT[key] = rec
            assert_map_close(res, want, key, rtol=MAP_RTOL if well else MAP_RTOL_CONTRAST)
        else:
            rec['criterion'] = 'strict %.0e' % MAP_RTOL
            PARITY_REPORT[key] = rec
            assert_map_close(res, want, key)
        if trace is not None and key.endswith('/ebp'):
            sums, names = trace
            assert_trace_close(sums, names, gold[key + '/trace'], gold[key + '/names'], key)
            g = np.asarray(gold[key + '/trace'], dtype=np.float64)
            n = min(len(g), len(sums))
            rec['trace_max_rel_err'] = float((np.abs(np.asarray(sums[:n], dtype=np.float64) - g[:n]) / np.maximum(np.abs(g[:n]), 1e-300)).max())
    return check


# ---- kernel-level ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('shape', [
    # cin, h, w, nb, cout, k, stride, pad
    (3, 32, 32, 3, 64, 7, 2, 3),       # stem-like: generic (ci,kh,kw) gather, K tail (147)
    (1, 20, 20, 2, 96, 5, 1, 2),       # Light-CNN conv1
    (64, 14, 14, 5, 64, 3, 1, 1),      # tap-major 3x3, ragged M (980)
    (48, 9, 9, 3, 96, 3, 1, 1),        # Cin % 32 != 0
    (256, 7, 7, 2, 130, 1, 1, 0),      # 1x1 float4 path needs M % 4 == 0: M = 98 -> dword path; ragged Cout
    (128, 8, 8, 4, 256, 1, 1, 0),      # 1x1 vector path
    (64, 16, 16, 2, 32, 1, 2, 0),      # strided 1x1
    (128, 8, 8, 2, 40, 8, 1, 0),       # Linear on a flattened 8x8 map (64 taps)
    (256, 14, 14, 5, 256, 3, 1, 1),    # layer-3 3x3: the bf16x6 kernel's gather (cfg 9), ragged M (980 = 7 x 128 + 84)
    (1024, 14, 14, 3, 128, 1, 1, 0),   # deep-K 1x1 on the bf16x6 kernel, one 128-row tile
    (8192, 1, 1, 2, 256, 1, 1, 0),     # a few tiles over a very deep K (the backward GEMM through a hooked classifier): up to 64 K-parts (cfg 0)
    (256, 1, 1, 1, 1037, 1, 1, 0),     # a classifier's forward for ONE image: ragged Cout (1037 = 16 x 64 + 13), one column
])
@pytest.mark.parametrize("cfg", [0, 4, 5, 6, 7, 8, 9, 10, 12, 10004, 20004, 30005, 80004, 20006, 30008, 20012])
def test_conv_gemm_matches_fp32_reference(gpu_device, shape, cfg):
    """The MFMA implicit-GEMM kernel against plain PyTorch fp32 conv2d on the CPU."""
    from xfr_amd import _lib
    lib = _lib.load()
    cin, h, w, nb, cout, k, stride, pad = shape
    g = torch.Generator().manual_seed(0)
    x = torch.randn((nb, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, k, k), generator=g) / np.sqrt(cin * k * k)
    b = torch.randn((cout,), generator=g)
    want = torch.nn.functional.conv2d(x, wt, b, stride=stride, padding=pad)
    xg = x.to(gpu_device).permute(1, 0, 2, 3).contiguous()
    out = torch.full((cout, nb) + tuple(want.shape[2:]), float('nan'), device=gpu_device)
    ms = ctypes.c_float()
    _lib.check(lib.xfr_debug_conv(xg.data_ptr(), wt.data_ptr(), b.data_ptr(), out.data_ptr(), cin, h, w, nb, cout, k, k,
                                  stride, pad, 0, cfg, 1, ctypes.byref(ms)))
    got = out.permute(1, 0, 2, 3).cpu()
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.parametrize('shape', [
    # cin, h, w, nb, cout, k, stride, pad
    (128, 28, 28, 3, 128, 3, 1, 1),    # patch staging, rows of 28 (halo 29), ragged M (2352 = 18 x 128 + 48), a single co tile
    (64, 56, 56, 1, 128, 3, 1, 1),     # halo 57 of the 64 a patch may take
    (512, 7, 7, 6, 256, 3, 1, 1),      # 7 x 7 images: a 128-column tile spans 2.6 images, most taps of a border pixel leave the image
    (32, 12, 12, 2, 128, 5, 1, 2),     # 5 x 5: 25 taps, not a multiple of three -> a KxK layer on the slab path
    (64, 10, 10, 3, 256, 3, 1, 0),     # a 'valid' 3 x 3 (pad 0): not a same-size layer -> slab path
    (256, 14, 14, 64, 256, 1, 1, 0),   # 1 x 1, K = 256 (the shallowest layer the engine sends), 2 x 98 tiles
    (16, 9, 9, 1, 128, 1, 1, 0),       # a single K step, M = 81: less than one tile
    (48, 8, 8, 2, 384, 3, 1, 1),       # three co tiles, three channel blocks
])
def test_bf16x6_kernel_shapes(gpu_device, shape):
    """conv_gemm_split_kernel (K17; cfg 9 of xfr_debug_conv runs it on every shape it CAN run, whatever the engine's depth / grid rules) at the edges of
    its two staging modes, against float64 conv2d: the fp32 operands are split exactly, so the bar is the fp32 kernels' (2e-5 of the maximum) and the
    kernel in fact stays far below it (asserted at 4e-6).  The launch counter proves the bf16x6 kernel ran, twice the same bits."""
    from xfr_amd import _lib
    from xfr_amd.engine import Engine
    lib = _lib.load()
    cin, h, w, nb, cout, k, stride, pad = shape
    bb, sd = make_backbone('stresnet_mini', seed=0, num_classes=3)
    eng = Engine(bb.build_program(), 2, gpu_device)        # (only for the process-wide launch counter's entry point)
    g = torch.Generator().manual_seed(2)
    x = torch.randn((nb, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, k, k), generator=g) / np.sqrt(cin * k * k)
    b = torch.randn((cout,), generator=g)
    want = torch.nn.functional.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=pad)
    xg = x.to(gpu_device).permute(1, 0, 2, 3).contiguous()
    outs = []
    for reps in (1, 3):
        out = torch.full((cout, nb) + tuple(want.shape[2:]), float('nan'), device=gpu_device)
        ms = ctypes.c_float()
        before = eng.split_gemm_launches()
        _lib.check(lib.xfr_debug_conv(xg.data_ptr(), wt.data_ptr(), b.data_ptr(), out.data_ptr(), cin, h, w, nb, cout, k, k,
                                      stride, pad, 0, 9, reps, ctypes.byref(ms)))
        assert eng.split_gemm_launches() - before == 2 * reps, 'the bf16x6 kernel did not run this shape'       # warm-up + timed launches
        outs.append(out.permute(1, 0, 2, 3).cpu())
    eng.close()
    assert torch.isfinite(outs[0]).all()
    assert float((outs[0].double() - want).abs().max()) <= 4e-6 * float(want.abs().max())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('shape', [
    (256, 14, 14, 64, 256, 3, 1, 1),   # ResNet-101 layer 3 at 64 images: 784 tiles = 3 x 256 + 16 tail tiles
    (1024, 14, 14, 32, 256, 1, 1, 0),  # 392 tiles: 136 tail tiles, float4 operand path
    (512, 7, 7, 32, 512, 3, 1, 1),     # 200 tiles: fewer tiles than CUs
])
@pytest.mark.parametrize('cfg', [0, 10004, 40004, 30005, 6, 40006, 20008, 30007])
def test_conv_gemm_tail_balancing(gpu_device, shape, cfg):
    """Whole tiles and K-parts of tail tiles in one grid (conv_gemm.hip, pick_tail_split): equal to fp32 conv2d, and
    launching repeatedly on the same scratch (arrival counters re-armed by the last part) gives the same bits."""
    from xfr_amd import _lib
    lib = _lib.load()
    cin, h, w, nb, cout, k, stride, pad = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn((nb, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, k, k), generator=g) / np.sqrt(cin * k * k)
    b = torch.randn((cout,), generator=g)
    want = torch.nn.functional.conv2d(x, wt, b, stride=stride, padding=pad)
    xg = x.to(gpu_device).permute(1, 0, 2, 3).contiguous()
    outs = []
    for reps in (1, 4):
        out = torch.full((cout, nb) + tuple(want.shape[2:]), float('nan'), device=gpu_device)
        ms = ctypes.c_float()
        _lib.check(lib.xfr_debug_conv(xg.data_ptr(), wt.data_ptr(), b.data_ptr(), out.data_ptr(), cin, h, w, nb, cout, k, k,
                                      stride, pad, 0, cfg, reps, ctypes.byref(ms)))
        outs.append(out.permute(1, 0, 2, 3).cpu())
    assert torch.isfinite(outs[0]).all()
    assert float((outs[0] - want).abs().max()) <= 2e-5 * float(want.abs().max())
    assert torch.equal(outs[0], outs[1])


def test_embeddings_and_mean_ebp(gpu_device):
    """Whitebox.embeddings (whitebox.py:747-785) and the mean-EBP prior call of generate_whitebox_saliency.py:207-214
    (P = ones over the hooked classifier)."""
    from oracle import ebp_oracle as O
    from xfr_amd.models import whitebox as WB
    bb, sd = make_backbone('stresnet_mini', seed=4, num_classes=11)
    x = make_images('stresnet_mini', 5, seed=2)
    wb = WB.Whitebox(WB.WhiteboxSTResnet(bb.to(gpu_device)))
    wb.batch_size = 2                                   # exercises the split into batches (:771)
    emb = wb.embeddings([xi for xi in x], norm=True)
    ow = O.OracleWhitebox('stresnet_mini', sd, ('hooked', None), 'affineonly_with_prior')
    want = ow.encode(x).numpy()
    want = want / np.linalg.norm(want, axis=1, keepdims=True)
    assert emb.shape == (5, 512) and np.abs(emb - want).max() <= 1e-5
    emb2 = wb.embeddings([xi.numpy() for xi in x], norm=False)
    assert np.abs(emb2 - ow.encode(x).numpy()).max() <= 1e-4 * np.abs(emb2).max()
    # file names / a DataFrame (whitebox.py:752-766 through image_loader + convert_from_numpy): the same encodings as the arrays they hold
    import tempfile
    import PIL.Image
    import pandas as pd
    rs = np.random.RandomState(5)
    pics = [rs.randint(0, 256, size=(224, 224, 3)).astype(np.uint8) for _ in range(3)]
    with tempfile.TemporaryDirectory() as d:
        fns = []
        for i, pic in enumerate(pics):
            fns.append(os.path.join(d, 'im%d.png' % i))
            PIL.Image.fromarray(pic).save(fns[-1])
        e_files = wb.embeddings(fns, norm=False)
        e_frame = wb.embeddings(pd.DataFrame([{'Filename': f, 'SubjectID': str(i)} for i, f in enumerate(fns)]), norm=False)
    xt = torch.cat([wb.convert_from_numpy(pic.astype(float) / 255) for pic in pics])
    e_want = ow.encode(xt).numpy()
    assert e_files.shape == (3, 512) and np.array_equal(e_files, e_frame)
    assert np.abs(e_files - e_want).max() <= 1e-4 * np.abs(e_want).max()
    ones = torch.ones((1, 11))
    got = wb.ebp(x[:1], ones)                            # mean EBP saliency
    assert_map_close_robust(got, ow.ebp(x[:1], ones), 'mean_ebp')


def test_saliency_blur_matches_scipy(gpu_device):
    from oracle.ebp_oracle import mwp_to_saliency
    from xfr_amd.models import whitebox as WB
    bb, _ = make_backbone('stresnet_mini')
    wb = WB.Whitebox(WB.WhiteboxSTResnet(bb.to(gpu_device)))
    rng = np.random.RandomState(1)
    for shape in ((112, 112), (128, 128), (17, 9)):
        P = (rng.rand(*shape) ** 4).astype(np.float32)
        got = wb._mwp_to_saliency(P)
        want = mwp_to_saliency(P)
        assert got.dtype == np.float32 and abs(float(got.sum()) - 1.0) < 1e-5
        assert np.abs(got - want).max() <= 2e-7 * want.max()
    z = wb._mwp_to_saliency(np.zeros((112, 112), np.float32))     # max(sum, eps) guard: whitebox.py:459
    assert np.all(z == 0)


# ---- golden vectors from the real reference -------------------------------------------------------------------------
@pytest.mark.parametrize('recipe', ['mild', 'harsh'])
@pytest.mark.parametrize('mode', ['affineonly_with_prior', 'norelu', 'all', 'affineonly'])
def test_mini_resnet_golden(gpu_device, recipe, mode):
    gold = GC.golden('golden_mini')
    bb, sd = make_backbone('stresnet_mini', seed=3, recipe=recipe, num_classes=5)
    assert synth.state_checksum(sd) == str(gold['mini/%s/wsum' % recipe])
    GC.replay(GC.engine_subject('stresnet_mini', bb, mode), GC.mini_cases(recipe, mode), gold, _check_factory())


@pytest.mark.parametrize('mode', ['affineonly_with_prior', 'norelu'])
def test_resnet101_demo_sequences_golden(gpu_device, mode):
    """demo/test_whitebox.py:77-144 call sequences on the bundled JPEGs, seeded weights."""
    gold = GC.golden('golden_r101')
    bb, sd = make_backbone('stresnet101', seed=0, num_classes=65359)
    assert synth.state_checksum(sd) == str(gold['r101/wsum'])
    subj = GC.engine_subject('stresnet101', bb, mode)
    x_demo, x_probe, x_non, x_mate = GC.net_inputs('stresnet101')
    enc = subj.enc(x_mate).cpu().numpy()
    assert np.abs(enc - gold['r101/%s/enc_mate' % mode]).max() <= 1e-4 * np.abs(enc).max()
    GC.replay(subj, GC.r101_cases(mode), gold, _check_factory())


@pytest.mark.parametrize('mode', ['affineonly_with_prior', 'norelu'])
def test_resnet50_128_golden(gpu_device, mode):
    gold = GC.golden('golden_r50')
    bb, sd = make_backbone('resnet50_128', seed=0)
    assert synth.state_checksum(sd) == str(gold['r50/wsum'])
    GC.replay(GC.engine_subject('resnet50_128', bb, mode), GC.r50_cases(mode), gold, _check_factory())


@pytest.mark.parametrize('mode', ['affineonly', 'affineonly_with_prior', 'all'])
def test_lightcnn_golden(gpu_device, mode):
    gold = GC.golden('golden_lcnn')
    bb, sd = make_backbone('lightcnn29v2', seed=0, num_classes=80013)
    assert synth.state_checksum(sd) == str(gold['lcnn/wsum'])
    GC.replay(GC.engine_subject('lightcnn29v2', bb, mode), GC.lcnn_cases(mode), gold, _check_factory())


@pytest.mark.parametrize('arch,tag,mode,nc', [('resnet50_128', 'r50', 'norelu', None), ('resnet50_128', 'r50', 'affineonly_with_prior', None),
                                               ('lightcnn29v2', 'lcnn', 'affineonly_with_prior', 7), ('lightcnn29v2', 'lcnn', 'all', 7)])
def test_well_conditioned_contrastive_golden(gpu_device, arch, tag, mode, nc):
    """Contrastive / truncated maps under two INDEPENDENT random classifier rows (tests/golden/make_golden_synth.py, the real reference): the
    contrast does not cancel, and the engine is held to 1e-3 of the map maximum instead of the 5e-3 the
    nearly-parallel face encodings of the demo cases need.  ResNet-101's case of the same kind is `r101/.../synthetic/contrastive`."""
    gold = GC.golden('golden_synth')
    bb, sd = make_backbone(arch, seed=0, num_classes=nc)
    assert synth.state_checksum(sd) == str(gold[tag + '/wsum'])
    GC.replay(GC.engine_subject(arch, bb, mode), GC.synth_cases(arch, tag, mode), gold, _check_factory())


@pytest.mark.parametrize('arch,mode', [('stresnet101', 'affineonly_with_prior'), ('stresnet101', 'norelu'), ('resnet50_128', 'norelu'),
                                       ('resnet50_128', 'affineonly_with_prior'), ('stresnet_mini', 'all')])
def test_golden_contrastive_cases_on_the_lean_schedule(gpu_device, arch, mode):
    """The contrastive / truncated golden cases of the BatchNorm backbones once more, as a batch of four copies of the probe: a multiple of four takes
    the lean schedule (stored hook quotients, one-bit gates; the replay above runs batches of one, i.e. the literal operands).  Same golden
    vectors, same tolerances; the dual-accumulator launches are counted."""
    which = ['contrastive', 'truncated']
    if arch == 'stresnet101':
        gold, cases = GC.golden('golden_r101'), GC.r101_cases(mode, which=which)
        bb, sd = make_backbone(arch, seed=0, num_classes=65359)
    elif arch == 'resnet50_128':
        gold = GC.golden('golden_r50')
        bb, sd = make_backbone(arch, seed=0)
        cases = GC.r50_cases(mode, which=which)
    else:
        gold, cases = GC.golden('golden_mini'), GC.mini_cases('mild', mode)
        cases = [c for c in cases if c[0] == '__set__' or c[0].endswith('contrastive') or c[0].endswith('truncated')]
        bb, sd = make_backbone(arch, seed=3, recipe='mild', num_classes=5)
    subj = GC.engine_subject(arch, bb, mode, replicate=True)
    inner = _check_factory()

    def check(key, res, trace, g):
        prev = PARITY_REPORT.get(key)            # the literal replay's entry of the same case, if that test ran before
        inner(key, res, None, g)
        PARITY_REPORT[key + ' [lean, batch of 4]'] = PARITY_REPORT.pop(key)
        if prev is not None:
            PARITY_REPORT[key] = prev
    GC.replay(subj, cases, gold, check)
    if arch == 'resnet50_128':
        gs = GC.golden('golden_synth')
        GC.replay(subj, GC.synth_cases(arch, 'r50', mode), gs, check)
    assert subj.wb._engine(4).lean_launches() > 0


@pytest.mark.parametrize('arch,mode', [('stresnet101', 'affineonly_with_prior'), ('stresnet101', 'norelu'), ('resnet50_128', 'norelu')])
def test_golden_cases_on_the_split_gemm(gpu_device, arch, mode):
    """xfr_engine_set_split_gemm(3 + 4): forward convolutions AND the sweep's backward-data GEMMs of the covered layers on the bf16 matrix pipe (bf16x6,
    conv_gemm_split.hip K17) -- the default mode since round 6, here for EVERY grid (+ 4): these replays run batches of one and four, whose launches the
    default grid rule leaves on the fp32 kernels.  The golden cases of the ResNets -- same vectors from the real reference, same tolerances; the kernel's
    launches are counted."""
    if arch == 'stresnet101':
        gold, cases = GC.golden('golden_r101'), GC.r101_cases(mode)
        bb, sd = make_backbone(arch, seed=0, num_classes=65359)
    else:
        gold, cases = GC.golden('golden_r50'), GC.r50_cases(mode)
        bb, sd = make_backbone(arch, seed=0)
    subj = GC.engine_subject(arch, bb, mode)
    eng = subj.wb._engine(1)
    eng.set_split_gemm(3 | 4)      # + 4: whatever the grid (these replays run batches of one and four; by default such launches stay on the fp32 kernels)
    before = eng.split_gemm_launches()
    inner = _check_factory()

    def check(key, res, trace, g):
        prev = PARITY_REPORT.get(key)
        inner(key, res, trace, g)
        PARITY_REPORT[key + ' [bf16x6]'] = PARITY_REPORT.pop(key)
        if prev is not None:
            PARITY_REPORT[key] = prev
    GC.replay(subj, cases, gold, check)
    if arch == 'resnet50_128':
        GC.replay(subj, GC.synth_cases(arch, 'r50', mode), GC.golden('golden_synth'), check)
    assert subj.wb._engine(1).split_gemm_launches() > before


@pytest.mark.parametrize('arch,mode', [('stresnet101', 'affineonly_with_prior'), ('resnet50_128', 'norelu')])
def test_split_gemm_equals_fp32_kernels(gpu_device, arch, mode):
    """The bf16x6 kernel (mode 1: forward convolutions; mode 3, the default: backward-data GEMMs too; both + 4 = every grid) against the fp32 MFMA kernels
    (mode 0) on the same engine, same inputs: every fp32 operand is the exact sum of three bf16 pieces, the six piece products of order <= 2 are exact in
    fp32 and partial sums of 48 K-terms are folded into fp32 registers; what differs is the summation order (rms error against float64 BELOW the fp32
    kernels', tools/conv_error_probe.py).  Encodings 1e-5, plain-EBP maps 5e-5 of the maximum, contrastive maps (a difference of nearly equal tensors:
    parity_utils) LEAN_RTOL_CONTRAST."""
    n = 4
    bb, sd = make_backbone(arch, seed=6, num_classes=None if arch == 'resnet50_128' else 7)
    subj = GC.engine_subject(arch, bb, mode)
    wb = subj.wb
    x = make_images(arch, n, seed=33, smooth=True).to(gpu_device)
    D = emb_dim(arch)
    xm = (synth.unit_rows(n, D, seed=3) / 2500).to(gpu_device)
    xn = (synth.unit_rows(n, D, seed=4) / 2500).to(gpu_device)
    subj.set_cls(xm[:1].cpu(), xn[:1].cpu())
    eng = wb._engine(2 * n)
    res, launches = {}, {}
    for split in (3, 1, 0):
        eng.set_split_gemm(split | 4 if split else 0)       # + 4: small grids too (four images; the default rule would leave them on the fp32 kernels)
        before = eng.split_gemm_launches()
        res[split] = (wb.encode(x).clone(), wb.contrastive_triplet_ebp_batch(x, xm, xn).clone(),
                      wb.contrastive_triplet_ebp_batch(x, xm, xn, percentile=20).clone(),
                      torch.as_tensor(wb.ebp(x, torch.tensor([[1.0, 0.0]]))))
        launches[split] = eng.split_gemm_launches() - before
    eng.set_split_gemm(3)
    assert launches[3] > launches[1] > 0 and launches[0] == 0, launches
    for split in (3, 1):
        e1, e0 = res[split][0].float().cpu(), res[0][0].float().cpu()
        assert float((e1 - e0).abs().max()) <= 1e-5 * float(e0.abs().max())
        for what, a, b in zip(('contrastive (triplet entry)', 'truncated (triplet entry)', 'ebp'), res[split][1:], res[0][1:]):
            a, b = a.float().cpu().numpy(), b.float().cpu().numpy()
            for i in range(a.shape[0]):
                tag = '%s/%s bf16x6 mode %d against fp32: %s row %d' % (arch, mode, split, what, i)
                rel, cos = map_metrics(a[i], b[i])
                PARITY_REPORT['split-vs-fp32/' + tag] = {'max_abs_diff_over_max': float(rel), 'cosine': float(cos), 'criterion': 'against the fp32 kernels'}
                if what == 'ebp':
                    assert_map_close(a[i], b[i], tag, rtol=5e-5)
                elif what.startswith('truncated'):
                    assert_map_close_robust(a[i], b[i], tag, rtol=LEAN_RTOL_CONTRAST)
                else:
                    assert_map_close(a[i], b[i], tag, rtol=LEAN_RTOL_CONTRAST)


def test_bf16_planes_follow_the_weights(gpu_device):
    """Round-5 verdict, weak 6: a caller keeps the arena view, runs a forward (the bf16x6 planes of the covered packs exist), writes NEW weights through
    the view and calls mark_weights_loaded.  The bf16x6 forward must then run on the new weights like the fp32 kernels do -- stale planes would leave the
    deep-K convolutions on the old ones, silently."""
    arch, mode = 'resnet50_128', 'norelu'
    bb, _ = make_backbone(arch, seed=6)
    bb2, _ = make_backbone(arch, seed=9)
    subj, subj2 = GC.engine_subject(arch, bb, mode), GC.engine_subject(arch, bb2, mode)
    x = make_images(arch, 2, seed=5, smooth=True).to(gpu_device)
    eng, eng2 = subj.wb._engine(4), subj2.wb._engine(4)
    enc = subj.wb.net._program.marks['encode']
    eng.set_split_gemm(3 | 4)
    view = eng.weight_arena()                       # the retained pointer
    before = eng.split_gemm_launches()
    old = eng.forward(x, enc).clone()               # planes are (re)built for this forward
    assert eng.split_gemm_launches() > before
    view.copy_(eng2.weight_arena())
    eng.mark_weights_loaded()
    got = eng.forward(x, enc).clone()
    eng.set_split_gemm(0)
    want = eng.forward(x, enc).clone()
    eng2.set_split_gemm(0)
    ref = eng2.forward(x, enc)
    assert torch.equal(want, ref)                   # the fp32 kernels read the arena itself
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 1e-5 * scale, 'bf16x6 forward ran on stale planes'
    assert float((old - want).abs().max()) > 1e-2 * scale      # (the two weight sets do differ)


# ---- oracle on fresh seeded inputs, batched ----------------------------------------------------------------------------
@pytest.mark.parametrize('arch,mode', [('stresnet_mini', 'affineonly_with_prior'), ('stresnet_mini', 'norelu'),
                                       ('lightcnn29v2', 'affineonly_with_prior'), ('resnet50_128', 'norelu')])
def test_batched_engine_equals_per_sample_oracle(gpu_device, arch, mode):
    """The reference is batch-1 for contrastive EBP (whitebox.py:512,524); the batched engine must equal it per sample."""
    from oracle import ebp_oracle as O
    from xfr_amd.models import whitebox as WB
    n = 3
    bb, sd = make_backbone(arch, seed=21, num_classes=None if arch == 'resnet50_128' else 9)
    x = make_images(arch, n, seed=77, smooth=False)
    D = emb_dim(arch)
    xm = synth.unit_rows(n, D, seed=3) / 2500
    xn = synth.unit_rows(n, D, seed=4) / 2500
    subj = GC.engine_subject(arch, bb, mode)
    # classifier rows are needed for the marks/engine to exist
    subj.set_cls(xm[:1], xn[:1])
    sal = subj.wb.contrastive_triplet_ebp_batch(x, xm, xn).cpu().numpy()
    assert sal.shape[0] == n
    for i in range(n):
        ow = O.OracleWhitebox(arch, sd, ('hooked', None), mode)
        ow.set_triplet_classifier(xm[i:i + 1], xn[i:i + 1])
        want = ow.contrastive_ebp(x[i:i + 1], 0, 1)
        assert_map_close_robust(sal[i], want, '%s %s sample %d' % (arch, mode, i))
        assert abs(float(sal[i].sum()) - 1.0) < 1e-4


def test_uint8_saliency_versions(gpu_device):
    """ebp_version 5 / 7-12 convert through uint8 + PIL blur on the host (whitebox.py:285,451-454)."""
    from oracle import ebp_oracle as O
    from xfr_amd.models import whitebox as WB
    bb, sd = make_backbone('stresnet_mini', seed=9, num_classes=5)
    x = make_images('stresnet_mini', 1, seed=11)
    wbn = WB.WhiteboxSTResnet(bb.to(gpu_device))
    wb = WB.Whitebox(wbn, ebp_version=5)
    ow = O.OracleWhitebox('stresnet_mini', sd, ('hooked', None), 'affineonly_with_prior', ebp_version=5)
    xm, xn = synth.unit_rows(1, 512, seed=1) / 2500, synth.unit_rows(1, 512, seed=2) / 2500
    wbn.set_triplet_classifier(xm, xn)
    ow.set_triplet_classifier(xm, xn)
    for got, want in ((wb.contrastive_ebp(x, 0, 1), ow.contrastive_ebp(x, 0, 1)),
                      (wb.truncated_contrastive_ebp(x, 0, 1, 20), ow.truncated_contrastive_ebp(x, 0, 1, 20)),
                      (wb.ebp(x, torch.tensor([[1.0, 0.0]])), ow.ebp(x, torch.tensor([[1.0, 0.0]])))):
        assert got.dtype == np.uint8 and got.shape == (112, 112)
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        # two uint8 quantisations with a min-max stretch in between: a +-1 level flip before the blur is stretched to a few
        # levels after it; the maps still agree on average to a fraction of a level
        assert d.max() <= 10 and d.mean() < 0.5 and (d > 2).mean() < 0.02, (d.max(), d.mean())


def test_with_bias_mode_vs_oracle(gpu_device):
    """Whitebox(with_bias=True) (ebp_version 11, whitebox.py:286-289,321-324): positive pass uses relu(bias), relu(beta)."""
    from oracle import ebp_oracle as O
    from xfr_amd.models import whitebox as WB
    bb, sd = make_backbone('stresnet_mini', seed=9, num_classes=5)
    x = make_images('stresnet_mini', 1, seed=11)
    bb.to(gpu_device)
    wb = WB.Whitebox(WB.WhiteboxSTResnet(bb), ebp_subtree_mode='all', with_bias=True)
    wb.debug_trace = True
    ow = O.OracleWhitebox('stresnet_mini', sd, ('hooked', None), 'all', with_bias=True)
    Pn = torch.zeros(1, 5)
    Pn[0, 3] = 1
    got = wb.ebp(x, Pn, mwp=True)
    want = ow.ebp(x, Pn, mwp=True)
    assert_map_close_robust(got, want, 'with_bias ebp')
    sums = np.array([float(p.double().sum()) for p in ow.P])
    assert_trace_close(np.asarray(wb.P_trace)[:, 0], wb.P_layername, sums, np.array(ow.P_layername), 'with_bias trace')
    # and it differs from with_bias=False (the flag is live)
    wb2 = WB.Whitebox(wb.net, ebp_subtree_mode='all', with_bias=False)
    other = wb2.ebp(x, Pn, mwp=True)
    assert np.abs(other - got).max() > 1e-6 * np.abs(got).max()
    # the lean schedule under with_bias (its dual-accumulator epilogue adds relu(bias) to the relu(W) tile, relu(beta) in the ReLU quotient): a batch
    # of four copies, un-traced, against the oracle and against the literal schedule
    wb.debug_trace = False
    eng = wb._engine(4)
    before = eng.lean_launches()
    lean = wb.ebp(x.repeat(4, 1, 1, 1), Pn, mwp=True)
    assert eng.lean_launches() > before
    assert_map_close_robust(lean[0], want, 'with_bias ebp, lean')
    eng.set_lean(False)
    lit = wb.ebp(x.repeat(4, 1, 1, 1), Pn, mwp=True)
    eng.set_lean(True)
    assert_map_close(lean[0], lit[0], 'with_bias lean vs literal', rtol=1e-5)


def test_truncation_tail_equals_reference_formula_on_engine_P(gpu_device):
    """whitebox.py:547-558 (sort, cumsum, percentile mask) applied on the CPU to the engine's own P[-2] must give the
    engine's truncated map: isolates the radix-select tail from the (discontinuous) dependence on P."""
    import torch.nn.functional as F
    from oracle.ebp_oracle import mwp_to_saliency
    bb, sd = make_backbone('stresnet_mini', seed=3, num_classes=7)
    x = make_images('stresnet_mini', 2)
    subj = GC.engine_subject('stresnet_mini', bb, 'norelu')
    subj.set_cls(synth.unit_rows(1, 512, seed=1) / 2500, synth.unit_rows(1, 512, seed=2) / 2500)
    wb = subj.wb
    eng = wb._engine(2)
    st, seeds = wb._class_seeds(2, 0, 1)
    mwp, _ = eng.ebp(x, st, seeds, want_mwp=True)
    mwp = mwp.cpu()
    for pct in (20.0, 50.0, 0.0):
        sal = eng.contrastive(x, st, seeds, pct).cpu().numpy()
        for i in range(2):
            m = mwp[0, i:i + 1] / torch.sum(mwp[0, i:i + 1])
            q = mwp[1, i:i + 1] / torch.sum(mwp[1, i:i + 1])
            (s, idx) = torch.sort(torch.flatten(m.clone()))
            cs = torch.cumsum(s, 0)
            mask = torch.zeros(s.shape)
            mask[idx] = (cs >= (pct / 100.0) * cs[-1]).type(torch.FloatTensor)
            mask = mask.reshape(m.shape)
            c = np.squeeze(np.sum(F.relu(mask * m - mask * q).numpy(), axis=1).astype(np.float32))
            assert_map_close(sal[i], mwp_to_saliency(c), 'truncated tail pct=%g sample %d' % (pct, i))


# ---- full BASELINE.json size: properties that do not need the (slow) CPU path --------------------------------------------
def test_resnet101_batch32_properties(gpu_device):
    """ResNet-101, 32 triplets (BASELINE.json configs[1]): every map is finite, non-negative and sums to 1; samples
    are independent (a sample computed alone gives the same map); the batch is permutation-equivariant.  With the three
    batch-dependent defaults off (GEMM tail balancing, the lean schedule, the bf16x6 kernel: include/xfr_amd.h, Conventions) the
    arithmetic is batch-invariant and both hold to fp32 noise of the final normalisation; with them on K is summed in a
    different order (cut tiles, another kernel for the small grid), so they hold to the contrastive tolerance
    (parity_utils.MAP_RTOL_CONTRAST)."""
    bb, sd = make_backbone('stresnet101', seed=0, num_classes=2)
    subj = GC.engine_subject('stresnet101', bb, 'affineonly_with_prior')
    wb = subj.wb
    B = 32
    imgs = make_images('stresnet101', B, seed=1234, smooth=False).to(gpu_device)
    # well-separated mate / non-mate directions: noise-image encodings are nearly parallel and the contrast then
    # amplifies one-ulp differences (parity section)
    em = (synth.unit_rows(B, 512, seed=5) / 2500).to(gpu_device)
    en = (synth.unit_rows(B, 512, seed=6) / 2500).to(gpu_device)
    subj.set_cls(em[:1].cpu(), en[:1].cpu())
    probes = imgs
    for balanced, tol in ((False, 1e-5), (True, MAP_RTOL_CONTRAST)):
        wb._engine(B).set_tail_balance(balanced)       # one engine serves every batch size up to B
        # batch-invariant arithmetic also needs ONE form of the hooks for every batch size: the lean schedule applies to batches that are a
        # multiple of four (32 yes, 1 no), so the strict leg runs the literal schedule; the default leg compares lean (32) with literal (1)
        wb._engine(B).set_lean(balanced)
        # ... and ONE kernel per layer: by default the bf16x6 kernel takes a covered layer's launch only when its grid is large enough (32 images yes, 1 no)
        wb._engine(B).set_split_gemm(3 if balanced else 0)
        sal = wb.contrastive_triplet_ebp_batch(probes, em, en)
        assert tuple(sal.shape) == (B, 112, 112)
        assert bool(torch.isfinite(sal).all()) and float(sal.min()) >= 0.0
        assert float((sal.sum(dim=(1, 2)) - 1.0).abs().max()) < 1e-4
        for i in (0, 13, 31):
            alone = wb.contrastive_triplet_ebp_batch(probes[i:i + 1], em[i:i + 1], en[i:i + 1])
            rel, cos = map_metrics(alone[0].cpu().numpy(), sal[i].cpu().numpy())
            assert rel <= tol and cos >= (0.9999999 if not balanced else 0.99999), (balanced, i, rel, cos)
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).to(gpu_device)
        sal_p = wb.contrastive_triplet_ebp_batch(probes[perm], em[perm], en[perm])
        assert float((sal_p - sal[perm]).abs().max()) <= tol * float(sal.max()), balanced
        again = wb.contrastive_triplet_ebp_batch(probes, em, en)
        assert torch.equal(again, sal)                 # run-to-run deterministic in both modes
    # truncated variant at full size: same invariants
    sal_t = wb.contrastive_triplet_ebp_batch(probes, em, en, percentile=20)
    assert bool(torch.isfinite(sal_t).all()) and float((sal_t.sum(dim=(1, 2)) - 1.0).abs().max()) < 1e-4


@pytest.mark.parametrize('arch,mode,B,pct', [('resnet50_128', 'norelu', 64, 20),           # BASELINE.json configs[2]
                                             ('lightcnn29v2', 'affineonly', 128, None)])     # BASELINE.json configs[3]
def test_secondary_configs_full_size_properties(gpu_device, arch, mode, B, pct):
    """The other two single-GPU configurations at their full batch: finite, non-negative, unit-sum maps, bit-identical
    between runs; a sample computed alone equals its row of the batch (batch-invariant arithmetic: 1e-5; default: the
    map tolerance); the first two samples also agree with the per-sample CPU oracle."""
    from oracle import ebp_oracle as O
    bb, sd = make_backbone(arch, seed=5, num_classes=None if arch == 'resnet50_128' else 80013)
    x = make_images(arch, B, seed=4321, smooth=False).to(gpu_device)
    subj = GC.engine_subject(arch, bb, mode)
    wb = subj.wb
    if arch == 'resnet50_128':
        D = emb_dim(arch)
        xm = (synth.unit_rows(B, D, seed=7) / 2500).to(gpu_device)
        xn = (synth.unit_rows(B, D, seed=8) / 2500).to(gpu_device)
        subj.set_cls(xm[:1].cpu(), xn[:1].cpu())
        run = lambda lo, hi: wb.contrastive_triplet_ebp_batch(x[lo:hi], xm[lo:hi], xn[lo:hi], percentile=pct)   # noqa: E731
        tol_on = MAP_RTOL_CONTRAST
    else:
        onehot = torch.zeros((1, 80013)); onehot[0, 0] = 1.0
        def run(lo, hi):                                   # Whitebox.ebp takes N x C seeds (whitebox.py:482-504)
            out = np.asarray(wb.ebp(x[lo:hi], onehot.expand(hi - lo, -1)))
            return torch.as_tensor(out.reshape((hi - lo,) + out.shape[-2:]))
        tol_on = 1e-3
    eng = wb._engine(B)
    for balanced, tol in ((False, 1e-5), (True, tol_on)):
        eng.set_tail_balance(balanced)
        sal = run(0, B)
        sal = sal if torch.is_tensor(sal) else torch.as_tensor(sal)
        assert sal.shape[0] == B and bool(torch.isfinite(sal).all()) and float(sal.min()) >= 0.0
        assert float((sal.sum(dim=(1, 2)) - 1.0).abs().max()) < 1e-4
        assert torch.equal(run(0, B).cpu(), sal.cpu())
        for i in (0, B // 2 + 1, B - 1):
            alone = run(i, i + 1)
            rel, cos = map_metrics(alone[0].cpu().numpy(), sal[i].cpu().numpy())
            assert rel <= tol and cos >= 0.99999, (balanced, i, rel, cos)
    for i in range(2):
        ow = O.OracleWhitebox(arch, sd, ('hooked', None), mode)
        if arch == 'resnet50_128':
            ow.set_triplet_classifier(xm[i:i + 1].cpu(), xn[i:i + 1].cpu())
            want = ow.truncated_contrastive_ebp(x[i:i + 1].cpu(), 0, 1, percentile=pct)
            assert_map_close_robust(sal[i].cpu().numpy(), want, '%s sample %d' % (arch, i), rtol=MAP_RTOL_CONTRAST)
        else:
            want = ow.ebp(x[i:i + 1].cpu(), onehot)
            assert_map_close_robust(sal[i].cpu().numpy(), want, '%s sample %d' % (arch, i))


def test_triplet_step_equals_two_call_path(gpu_device):
    """xfr_triplet_contrastive (fused + two-stream) == encode(), encode(), contrastive() done call by call."""
    bb, sd = make_backbone('stresnet_mini', seed=3, num_classes=5)
    subj = GC.engine_subject('stresnet_mini', bb, 'affineonly_with_prior')
    wb = subj.wb
    n = 4
    imgs = make_images('stresnet_mini', 3 * n, seed=9, smooth=False).to(gpu_device)
    mates, nonmates, probes = imgs[:n], imgs[n:2 * n], imgs[2 * n:]
    # the two paths run the encodes at different batch sizes: compare them in batch-invariant arithmetic
    wb._engine(2 * n).set_tail_balance(False)
    for pct in (None, 20):
        got = wb.triplet_images_ebp_batch(probes, mates, nonmates, percentile=pct)
        em = (1.0 / 2500.0) * subj.enc(mates)
        en = (1.0 / 2500.0) * subj.enc(nonmates)
        want = wb.contrastive_triplet_ebp_batch(probes, em, en, percentile=pct)
        assert float((got - want).abs().max()) <= 1e-6 * float(want.max())
        again = wb.triplet_images_ebp_batch(probes, mates, nonmates, percentile=pct)
        assert torch.equal(got, again)        # deterministic, and the stream join is race-free
    # cross-call pipelining: a stream of different batches gives the same maps as one-by-one execution
    eng = wb._engine(2 * n)
    batches = [(probes, mates, nonmates), (mates, nonmates, probes), (nonmates, probes, mates), (probes, nonmates, mates)]
    ref = [wb.triplet_images_ebp_batch(*b).clone() for b in batches]
    gal = [torch.cat((b[1], b[2]), dim=0) for b in batches]
    torch.cuda.synchronize()                       # inputs_ready contract: inputs valid before the calls
    eng.set_pipeline(True)
    outs = [wb.triplet_images_ebp_batch(b[0], None, None, gallery=g, inputs_ready=True) for b, g in list(zip(batches, gal)) * 2]
    outs += [wb.triplet_images_ebp_batch(*b) for b in batches]      # inputs produced on the stream: still correct
    torch.cuda.synchronize()
    eng.set_pipeline(False)
    for i, o in enumerate(outs):
        assert torch.equal(o, ref[i % len(batches)]), 'pipelined call %d differs' % i


@pytest.mark.parametrize('level', [2, 6])
def test_pipelined_plain_calls(gpu_device, level):
    """xfr_engine_set_pipeline(2): ebp / contrastive calls overlap across calls and still give the one-by-one results; level 6 = the same with
    three forward slots (bit 2: the forwards may run two calls ahead of the sweep)."""
    bb, sd = make_backbone('lightcnn29v2', seed=2, num_classes=7)
    subj = GC.engine_subject('lightcnn29v2', bb, 'affineonly_with_prior')
    subj.wb.debug_trace = False
    wb = subj.wb
    eng = wb._engine(4)
    xs = [make_images('lightcnn29v2', 3, seed=10 + i, smooth=False).to(gpu_device) for i in range(4)]
    Pn = torch.zeros((1, 7)); Pn[0, 2] = 1
    st, seed = wb.net.seed_for(Pn, 3)
    seed = seed.unsqueeze(0).contiguous()
    ref = [eng.ebp(x, st, seed)[1].clone() for x in xs]
    enc = [eng.forward(x, wb.net._program.marks['encode']).clone() for x in xs]
    torch.cuda.synchronize()
    eng.set_pipeline(level)
    outs = [eng.ebp(x, st, seed)[1] for x in xs * 2]
    mid = eng.forward(xs[1], wb.net._program.marks['encode'])          # un-pipelined call in between
    outs2 = [eng.ebp(x, st, seed)[1] for x in xs]
    torch.cuda.synchronize()
    eng.set_pipeline(0)
    for i, o in enumerate(outs + outs2):
        assert torch.equal(o, ref[i % 4]), i
    assert torch.equal(mid, enc[1])


def test_weight_arena_is_a_zero_copy_view(gpu_device):
    """The multi-GPU path broadcasts INTO the arena tensor: it must alias the engine's memory, not copy it."""
    import ctypes
    from xfr_amd import _lib
    bb, sd = make_backbone('stresnet_mini', seed=3, num_classes=5)
    subj = GC.engine_subject('stresnet_mini', bb, 'affineonly_with_prior')
    eng = subj.wb._engine(1)
    arena = eng.weight_arena()
    p, nbytes = ctypes.c_void_p(), ctypes.c_size_t()
    _lib.check(eng.lib.xfr_engine_weight_arena(eng._h, ctypes.byref(p), ctypes.byref(nbytes)))
    assert arena.data_ptr() == p.value and arena.numel() == nbytes.value and arena.dtype == torch.uint8
    # a second engine that only RECEIVES the arena (rank != 0) computes the same maps
    from xfr_amd.engine import Engine
    x = make_images('stresnet_mini', 2).to(gpu_device)
    want = eng.forward(x, subj.wb.net._program.marks['encode'])
    e2 = Engine(subj.wb.net._program, 32, gpu_device)
    e2.weight_arena().copy_(arena)
    e2.mark_weights_loaded()
    got = e2.forward(x, subj.wb.net._program.marks['encode'])
    assert torch.equal(got, want)


def test_engine_argument_errors(gpu_device):
    bb, _ = make_backbone('stresnet_mini', num_classes=5)
    subj = GC.engine_subject('stresnet_mini', bb, 'affineonly_with_prior')
    wb = subj.wb
    x = make_images('stresnet_mini', 1)
    with pytest.raises(AssertionError):                # whitebox.py:508-509 channel range asserts
        wb.contrastive_ebp(x, 0, 5)
    with pytest.raises(ValueError):
        subj.enc(torch.zeros(1, 3, 100, 100))
    with pytest.raises(ValueError):                    # empty batch
        subj.enc(torch.zeros(0, 3, 224, 224))
    with pytest.raises(ValueError):                    # more images than the engine was built for
        wb._engine(1).forward(torch.zeros(wb._engine(1).max_batch + 1, 3, 224, 224), 1)
    with pytest.raises(ValueError):
        wb._engine(1).ebp(x, 2, torch.zeros(3, 1, 1))  # n_streams / seed shape
    with pytest.raises(ValueError):
        wb.net.engine().set_mode('nonsense')


def test_hold_forward_shares_and_drops_state(gpu_device):
    """xfr_engine_hold_forward: calls on the same input skip the forward and give the same bits; a different input, a
    mode change or hold = 0 never reuses stale activations."""
    bb, sd = make_backbone('stresnet_mini', seed=3, num_classes=5)
    subj = GC.engine_subject('stresnet_mini', bb, 'affineonly_with_prior')
    wb = subj.wb
    subj.set_cls(synth.unit_rows(1, 512, seed=1) / 2500, synth.unit_rows(1, 512, seed=2) / 2500)
    x = make_images('stresnet_mini', 2, seed=5, smooth=False).to(gpu_device)
    a, b = x[0:1].contiguous(), x[1:2].contiguous()
    ref_a = wb.contrastive_ebp(a, 0, 1).copy()
    ref_b = wb.contrastive_ebp(b, 0, 1).copy()
    ref_t = wb.truncated_contrastive_ebp(a, 0, 1, percentile=20).copy()
    eng = wb._engine(1)
    eng.hold_forward(True)
    try:
        assert np.array_equal(wb.contrastive_ebp(a, 0, 1), ref_a)
        assert np.array_equal(wb.contrastive_ebp(a, 0, 1), ref_a)              # forward skipped
        assert np.array_equal(wb.truncated_contrastive_ebp(a, 0, 1, percentile=20), ref_t)
        assert np.array_equal(wb.contrastive_ebp(b, 0, 1), ref_b)              # other input: new forward
        assert np.array_equal(wb.contrastive_ebp(a, 0, 1), ref_a)
        enc = wb.encode(b)                                                      # another call kind in between
        assert np.array_equal(wb.contrastive_ebp(a, 0, 1), ref_a)
        assert torch.equal(enc, wb.encode(b))
    finally:
        eng.hold_forward(False)
    assert np.array_equal(wb.contrastive_ebp(b, 0, 1), ref_b)


@pytest.mark.parametrize('arch,mode', [('stresnet_mini', 'affineonly_with_prior'), ('stresnet_mini', 'norelu'), ('resnet50_128', 'norelu'),
                                       ('lightcnn29v2', 'affineonly')])
def test_epilogue_fusion_is_bit_identical(gpu_device, arch, mode):
    """xfr_engine_set_epilogue_fusion: hook chains / BatchNorm+add+ReLU inside the GEMM epilogue (default) or as their own
    launches run the same arithmetic in the same order: identical bits for encodings, EBP and contrastive maps."""
    bb, sd = make_backbone(arch, seed=6, num_classes=None if arch == 'resnet50_128' else 7)
    subj = GC.engine_subject(arch, bb, mode)
    wb = subj.wb
    n = 4
    x = make_images(arch, n, seed=31, smooth=False).to(gpu_device)
    D = emb_dim(arch)
    xm = (synth.unit_rows(n, D, seed=3) / 2500).to(gpu_device)
    xn = (synth.unit_rows(n, D, seed=4) / 2500).to(gpu_device)
    subj.set_cls(xm[:1].cpu(), xn[:1].cpu())
    eng = wb._engine(2 * n)
    eng.set_lean(False)        # the fusion levels re-arrange the LITERAL arithmetic (the lean schedule only exists at level 3: test_lean_schedule_equals_literal)
    res = {}
    # default; everything un-fused; plus the probe forward's BatchNorm / ReLU (FORK / STORE chains); the interpreted epilogue
    # ... and (11) the default with Light-CNN's pool pairs as separate kernels / VJP launches (0 has them separate too, on the un-fused schedule)
    # (19) the default with Light-CNN's first layer through the GEMM instead of the direct convolution
    # (35) the default without the stream-interleaved tile order of the two-stream backward GEMMs
    # (67) the default with the down-sampling blocks' shortcut VJP as separate launches (no EW_AVGUP_IN head, scattering GEMM)
    # (131) the default with the main path's chain of a projection-shortcut block as its own launch (no side branch in the GEMM epilogue)
    # (259) the default with the down-sampling blocks' shortcut computed in program order (behind the main path), their residual add as its own launch
    for fused in (1, 0, 3, 5, 11, 19, 35, 67, 131, 259):
        eng.set_epilogue_fusion(fused)
        res[fused] = (wb.encode(x).clone(), wb.contrastive_triplet_ebp_batch(x, xm, xn).clone(),
                      wb.contrastive_triplet_ebp_batch(x, xm, xn, percentile=20).clone(),
                      wb.triplet_images_ebp_batch(x[:2], x[2:4], x[1:3]).clone())
    eng.set_epilogue_fusion(3)
    eng.set_lean(True)
    for level in (0, 3, 5, 11, 19, 35, 67, 131, 259):
        for a, b in zip(res[1], res[level]):
            assert torch.equal(a, b), level


@pytest.mark.parametrize('mode', ['affineonly', 'affineonly_with_prior', 'all'])
def test_maxfeaturemap_net_outside_the_signature_table(gpu_device, mode):
    """A Conv -> Split -> max network whose merged fan-out chain is NOT in chain_sigs.inc (a Multiply between two MaxFeatureMap layers: the
    backward GEMM's chain is [conv hook, Multiply VJP, Multiply hook, MAXHALF fan-out, Split hook]).  The fan-out only exists as a compiled
    epilogue; round 3 launched it anyway and every ebp call of such a net failed with XFR_STATE_ERROR.  Now the plan falls back to the schedule
    without fan-outs (interpreted chain behind the GEMM, the MaxFeatureMap VJP as the head of its own chain launch): the maps must equal the
    CPU oracle's (a tape evaluated op by op)."""
    from oracle import ebp_oracle as O
    from xfr_amd.engine import Engine
    from xfr_amd.program import Program
    g = torch.Generator().manual_seed(11)
    H = 16
    sd = {'c1.weight': torch.randn((8, 1, 3, 3), generator=g) * 0.4, 'c1.bias': torch.randn((8,), generator=g) * 0.1,
          'c2.weight': torch.randn((8, 4, 3, 3), generator=g) * 0.2, 'c2.bias': torch.randn((8,), generator=g) * 0.1,
          'fc.weight': torch.randn((5, 4 * H * H), generator=g) * 0.05, 'fc.bias': torch.randn((5,), generator=g) * 0.1}
    prog = Program((1, H, H))
    t = prog.g_maxhalves(prog.split(prog.conv(0, 'c1', 8, 3, pad=1)))
    t = prog.multiply(t, 3.0)
    t = prog.g_maxhalves(prog.split(prog.conv(t, 'c2', 8, 3, pad=1)))
    t = prog.mark('classify', prog.linear(t, 'fc', 5, (H, H)))
    text = prog.describe(mode, t, batch=4)
    fan = [ln for ln in text.splitlines() if ln.startswith('bwd CONV_BWD') and ' SIG' in ln and any(c.endswith('d') and len(c) == 4 for c in ln.split(' SIG')[1].split()[:-1])]
    assert any('compiled=-1' in ln for ln in fan), text          # the planner WOULD fan out here, and no compiled epilogue exists for that chain

    n = 4
    x = torch.rand((n, 1, H, H), generator=g)
    Pn = torch.zeros((n, 5))
    Pn[:, 2] = 1.0
    eng = Engine(prog, n, gpu_device)
    eng.load_weights(sd)
    eng.set_mode(mode)
    _, pooled = eng.ebp(x.to(gpu_device), t, Pn.unsqueeze(0).to(gpu_device), want_mwp=False, want_pooled=True)
    got = pooled[0].cpu().numpy()

    def forward(tape, xx):
        u = tape.g_max_halves(tape.split(tape.conv(tape.input(xx), 'c1', stride=1, pad=1)))
        u = tape.multiply(u, 3.0)
        u = tape.g_max_halves(tape.split(tape.conv(u, 'c2', stride=1, pad=1)))
        return tape.linear(tape.g_flatten(u), 'fc')
    for i in range(n):
        tape = O.Tape({k: v.clone() for k, v in sd.items()})
        out = forward(tape, x[i:i + 1])
        P, names = tape.backward(out, Pn[i:i + 1], mode, 1e-16)
        want = P[-2].sum(dim=1)[0].numpy()          # MWP at the first convolution's output, channel-pooled (whitebox.py:499)
        assert_map_close(got[i], want, 'out-of-table MaxFeatureMap net, sample %d, %s' % (i, mode))
    eng.close()


@pytest.mark.parametrize('arch,mode,n', [('stresnet101', 'affineonly_with_prior', 8), ('stresnet101', 'norelu', 4), ('stresnet101', 'affineonly', 4),
                                         ('resnet50_128', 'norelu', 8), ('resnet50_128', 'affineonly_with_prior', 4), ('resnet50_128', 'all', 4),
                                         ('stresnet_mini', 'all', 4), ('lightcnn29v2', 'affineonly', 4)])
def test_lean_schedule_equals_literal(gpu_device, arch, mode, n):
    """xfr_engine_set_lean (the default): an un-observed sweep reads a stored quotient a / (x + eps) at the BatchNorm (and dividing ReLU) hooks
    and a one-bit gate at every hook whose x is its a, instead of the literal operands of whitebox.py:388-428.  Per hook that is at most one ulp
    away from the literal expression; here: encodings identical, contrastive / truncated / plain-EBP maps within 1e-5 of the map maximum of
    the literal schedule (the golden suite runs lean and holds both to the reference at its stated tolerances).  The ResNets really take it
    (dual-accumulator launches counted), a batch that is not a multiple of four and Light-CNN (no BatchNorm) stay literal."""
    bb, sd = make_backbone(arch, seed=6, num_classes=None if arch == 'resnet50_128' else 7)
    subj = GC.engine_subject(arch, bb, mode)
    wb = subj.wb
    x = make_images(arch, n, seed=33, smooth=True).to(gpu_device)
    D = emb_dim(arch)
    xm = (synth.unit_rows(n, D, seed=3) / 2500).to(gpu_device)
    xn = (synth.unit_rows(n, D, seed=4) / 2500).to(gpu_device)
    subj.set_cls(xm[:1].cpu(), xn[:1].cpu())
    eng = wb._engine(2 * n)
    eng.set_split_gemm(0)          # the lean transformation by itself: with bf16x6 forward convolutions (the default) the literal probe forward of a
                                   # K = 1152 layer is a bf16x6 dual launch where the lean one is the fp32 two-accumulator launch -- another 2e-5
    res, launches = {}, {}
    for lean in (1, 0):
        eng.set_lean(lean)
        before = eng.lean_launches()
        res[lean] = (wb.encode(x).clone(), wb.contrastive_triplet_ebp_batch(x, xm, xn).clone(),
                     wb.contrastive_triplet_ebp_batch(x, xm, xn, percentile=20).clone(),
                     torch.as_tensor(wb.contrastive_ebp(x, 0, 1)), torch.as_tensor(wb.ebp(x[:4], torch.tensor([[1.0, 0.0]]))))
        if arch.startswith('stresnet'):
            # ... and mean EBP over the HOOKED N-way classifier (generate_whitebox_saliency.py:207-214), whose Linear hook is a gate too
            saved = wb.net._classifier
            wb.net._classifier = None
            try:
                res[lean] = res[lean] + (torch.as_tensor(wb.ebp(x[:4], torch.ones((1, wb.net.num_classes())))),)
            finally:
                wb.net._classifier = saved
        launches[lean] = eng.lean_launches() - before
    assert launches[0] == 0
    assert (launches[1] > 0) == (arch != 'lightcnn29v2'), launches
    assert torch.equal(res[1][0], res[0][0])
    # Plain EBP maps: 1e-5 of the maximum.  Contrastive maps are a difference of two nearly equal normalised MWP tensors under these seeded
    # weights -- a last-bit change of P shows at 1e-4 .. 1e-3 of the map (parity_utils: the reference moves its OWN map by 5e-4 when one classifier
    # row moves by an ulp) -- so the lean and the literal sweep, one ulp apart per hook, are held to 2e-3 (LEAN_RTOL_CONTRAST; the golden replays hold BOTH to the reference) and to the
    # cosine bar; the truncated map's percentile mask may flip a handful of pixels on top (robust criterion).
    names = ('contrastive (triplet entry)', 'truncated (triplet entry)', 'contrastive_ebp', 'ebp', 'ebp')      # (the last: hooked classifier, ResNets only)
    for what, a, b in zip(names, res[1][1:], res[0][1:]):
        a, b = a.float().cpu().numpy(), b.float().cpu().numpy()
        for i in range(a.shape[0]):
            tag = '%s/%s %s row %d' % (arch, mode, what, i)
            if what == 'ebp':
                assert_map_close(a[i], b[i], tag, rtol=1e-5)
            elif what.startswith('truncated'):
                assert_map_close_robust(a[i], b[i], tag, rtol=LEAN_RTOL_CONTRAST)
            else:
                assert_map_close(a[i], b[i], tag, rtol=LEAN_RTOL_CONTRAST)
    # three probes: not a multiple of four -> the literal schedule, bit for bit the lean-off result
    eng.set_lean(1)
    before = eng.lean_launches()
    odd = wb.contrastive_triplet_ebp_batch(x[:3], xm[:3], xn[:3]).clone()
    assert eng.lean_launches() == before
    eng.set_lean(0)
    assert torch.equal(odd, wb.contrastive_triplet_ebp_batch(x[:3], xm[:3], xn[:3]))
    eng.set_lean(1)

