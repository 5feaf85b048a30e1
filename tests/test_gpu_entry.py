"""GPU tests of the entry points exactly as bench.py and the multi-GPU tools drive them (run with -m gpu on an MI355X):

* xfr_triplet_contrastive with cross-call pipelining and resident inputs at the full BASELINE.json batch sizes (ResNet-101
  B=32, ResNet-50-128d B=64 truncated), rows checked against the golden triplet of the reference and against the per-sample
  CPU oracle;
* the compiled chain epilogues cover every fused launch of those runs;
* Whitebox.P / P_layername / negative layer indices;
* two ranks (two processes on this one GPU, gloo) running the sharded inpainting-game workload and bench.py;
* the RCCL entry points of the C ABI at world size 1.
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import golden_cases as GC
from parity_utils import (MAP_RTOL_CONTRAST, assert_map_close, assert_map_close_robust, emb_dim, make_backbone, make_images,
                          map_metrics)
from xfr_amd import _lib, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _chain_stats():
    c, i, n = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32()
    _lib.check(_lib.load().xfr_chain_epilogue_stats(ctypes.byref(c), ctypes.byref(i), ctypes.byref(n)))
    return c.value, i.value, n.value


def _triplet_batch(arch, B, seed):
    """B triplets: row 0 is the reference's golden JPEG triplet (probe, mate, non-mate), the other rows are the same three
    photographs plus seeded smooth perturbations, so every row is a distinct, natural-image-like triplet."""
    x_demo, x_probe, x_non, x_mate = GC.net_inputs(arch)
    pert = make_images(arch, 3 * B, seed=seed, smooth=True)
    scale = torch.linspace(0.0, 0.35, B).reshape(B, 1, 1, 1)
    probes = x_probe + scale * pert[:B]
    mates = x_mate + scale * pert[B:2 * B]
    nonmates = x_non + scale * pert[2 * B:]
    return probes.contiguous(), mates.contiguous(), nonmates.contiguous()


@pytest.mark.parametrize('arch,mode,B,pct,gkey', [
    ('stresnet101', 'affineonly_with_prior', 32, None, 'r101/affineonly_with_prior/triplet/contrastive'),   # BASELINE.json configs[1] = bench.py
    ('resnet50_128', 'norelu', 64, 20, 'r50/norelu/triplet/truncated'),                                     # configs[2] = bench.py --model resnet50_128
])
def test_benchmarked_entry_point_full_batch(gpu_device, arch, mode, B, pct, gkey):
    """The call bench.py times -- xfr_triplet_contrastive, set_pipeline(1), inputs_ready=1, full batch -- checked row by row.

    The step computes its own classifier rows (the gallery forward), so the check is split the way the algorithm is:
    (a) the gallery encodings of row 0 equal the reference's golden encodings (1e-4);
    (b) every checked row equals the per-sample CPU oracle run with THAT row's classifier (contrastive tolerance): the whole
        EBP half -- probe forward with W and relu(W), fused hook-chain epilogues, two gradient streams, tail;
    (c) row 0 against the reference's golden map, classifier rows from the engine's own gallery forward (contrastive
        tolerance; mate / non-mate encodings under seeded random weights are nearly parallel -- cosine printed below -- which
        is what that tolerance is stated for);
    (d) calls pipelined across different batches return what the same calls return one by one, bit for bit."""
    from oracle import ebp_oracle as O
    from xfr_amd.engine import Engine
    gold = GC.golden('golden_r101' if arch == 'stresnet101' else 'golden_r50')
    bb, sd = make_backbone(arch, seed=0, num_classes=65359 if arch == 'stresnet101' else None)
    prog = bb.build_program()
    eng = Engine(prog, 2 * B, gpu_device)
    eng.load_weights(sd)
    eng.set_mode(mode)
    enc_t = prog.marks['encode']
    probes, mates, nonmates = (t.to(gpu_device) for t in _triplet_batch(arch, B, seed=77))
    gallery = torch.cat((mates, nonmates), dim=0)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).to(gpu_device)
    probes2, gallery2 = probes[perm].contiguous(), torch.cat((mates[perm], nonmates[perm]), dim=0)
    torch.cuda.synchronize()
    c0, i0, nsig = _chain_stats()
    # one by one (no pipelining) ...
    ref1 = eng.triplet_contrastive(probes, gallery, enc_t, 1.0 / 2500.0, pct).clone()
    ref2 = eng.triplet_contrastive(probes2, gallery2, enc_t, 1.0 / 2500.0, pct).clone()
    torch.cuda.synchronize()
    # ... and as bench.py runs them: pipelined, resident inputs
    eng.set_pipeline(1)
    outs = [eng.triplet_contrastive(p, g, enc_t, 1.0 / 2500.0, pct, inputs_ready=True) for p, g in
            ((probes, gallery), (probes2, gallery2), (probes, gallery), (probes2, gallery2))]
    torch.cuda.synchronize()
    eng.set_pipeline(0)
    for o, r in zip(outs, (ref1, ref2, ref1, ref2)):
        assert torch.equal(o, r)                                                     # (d)
    assert torch.equal(ref2, ref1[perm]) or float((ref2 - ref1[perm]).abs().max()) <= MAP_RTOL_CONTRAST * float(ref1.max())
    c1, i1, _ = _chain_stats()
    assert nsig > 0 and c1 > c0 and i1 == i0, 'a fused chain of %s ran through the interpreted epilogue' % arch
    sal = ref1.cpu().numpy()
    assert np.isfinite(sal).all() and sal.min() >= 0 and np.abs(sal.sum(axis=(1, 2)) - 1).max() < 1e-4
    # (a)
    enc = eng.forward(gallery, enc_t).reshape(2 * B, -1).cpu()
    pre = gkey.rsplit('/', 2)[0]
    gm, gn = torch.from_numpy(gold[pre + '/enc_mate']), torch.from_numpy(gold[pre + '/enc_nonmate'])
    assert float((enc[0:1] - gm).abs().max()) <= 1e-4 * float(gm.abs().max())
    assert float((enc[B:B + 1] - gn).abs().max()) <= 1e-4 * float(gn.abs().max())
    # (b)
    torch.set_num_threads(16)
    for i in (0, B // 2 - 3, B - 1):
        ow = O.OracleWhitebox(arch, sd, ('hooked', None), mode)
        ow.set_triplet_classifier(enc[i:i + 1] / 2500.0, enc[B + i:B + i + 1] / 2500.0)
        x = probes[i:i + 1].cpu()
        want = ow.contrastive_ebp(x, 0, 1) if pct is None else ow.truncated_contrastive_ebp(x, 0, 1, percentile=pct)
        if pct is None:
            assert_map_close(sal[i], want, '%s row %d vs oracle' % (arch, i), rtol=MAP_RTOL_CONTRAST)
        else:
            assert_map_close_robust(sal[i], want, '%s row %d vs oracle' % (arch, i), rtol=MAP_RTOL_CONTRAST)
    # (c)
    rel, cos = map_metrics(sal[0], gold[gkey + '/map'])
    cmn = float(torch.nn.functional.cosine_similarity(gm, gn).item())
    print('%s row 0 vs golden: max|d|/max %.2e, cosine %.7f (cosine(mate, non-mate) = %.5f)' % (arch, rel, cos, cmn))
    # measured on MI355X: 1.0e-3 / 0.9999998 (ResNet-101), 2.7e-4 / 1.0000000 (ResNet-50-128d)
    (assert_map_close if pct is None else assert_map_close_robust)(sal[0], gold[gkey + '/map'], '%s row 0 vs golden' % arch, rtol=MAP_RTOL_CONTRAST)
    eng.close()


def test_whitebox_P_surface(gpu_device):
    """Whitebox.P[k] for any k (whitebox.py:394), P_layername, len(P) and Python-style negative layer indices in
    layerwise_ebp (:570-577) against the oracle's full P list."""
    from oracle import ebp_oracle as O
    from xfr_amd.models import whitebox as WB
    bb, sd = make_backbone('stresnet_mini', seed=3, num_classes=5)
    x = make_images('stresnet_mini', 1, seed=5)
    wb = WB.Whitebox(WB.WhiteboxSTResnet(bb.to(gpu_device)), ebp_subtree_mode='norelu')
    ow = O.OracleWhitebox('stresnet_mini', sd, ('hooked', None), 'norelu')
    Pn = torch.zeros(1, 5)
    Pn[0, 2] = 1
    wb.ebp(x, Pn)
    ow.ebp(x, Pn)
    # P_layername: str(module) per entry of P, image hook included (whitebox.py:393); the oracle's list holds the class names
    assert len(wb.P) == len(ow.P) == len(wb.P_layername)
    assert [n.split('(')[0] for n in wb.P_layername] == [n.split('(')[0] for n in ow.P_layername]
    assert wb.P_layername[-1].startswith('Conv2d(3, 64, kernel_size=(7, 7)') or wb.P_layername[-1].startswith('Conv2d(')
    for k in (0, 7, 23, len(ow.P) - 2, -2, -5, -1):           # -1: the MWP at the image (the first layer's backward-data pass, on demand)
        got, want = wb.P[k].cpu().numpy(), ow.P[k].numpy()
        assert got.shape == want.shape
        assert np.abs(want).max() > 0
        assert np.abs(got - want).max() <= 1e-4 * max(np.abs(want).max(), 1e-30), k
    with pytest.raises(IndexError):
        wb.P[len(ow.P)]
    nf = len(ow.P) - 1
    for k in (-3, -25):                                                       # Python indexing into the nf + 1 entries of P
        a = wb.layerwise_ebp(x, k_layer=k, mode='argmax', k_poschannel=2)
        b = ow.layerwise_ebp(x, k_layer=k, mode='argmax', k_poschannel=2)
        if np.abs(b).max() == 0:
            assert np.abs(a).max() == 0
        else:
            assert_map_close_robust(a, b, 'layerwise k=%d' % k)
    pos = wb.layerwise_ebp(x, k_layer=20, mode='argmax', k_poschannel=2)           # a layer whose subtree reaches the image
    neg = wb.layerwise_ebp(x, k_layer=20 - (nf + 1), mode='argmax', k_poschannel=2)
    assert np.abs(pos).max() > 0 and np.array_equal(pos, neg)
    assert_map_close_robust(pos, ow.layerwise_ebp(x, k_layer=20, mode='argmax', k_poschannel=2), 'layerwise k=20')
    assert np.all(wb.layerwise_ebp(x, k_layer=nf, mode='argmax', k_poschannel=2) == 0)       # the image hook: nothing reaches P[-2]
    assert np.all(ow.layerwise_ebp(x, k_layer=nf, mode='argmax', k_poschannel=2) == 0)
    with pytest.raises(IndexError):
        wb.layerwise_ebp(x, k_layer=nf + 1)


def test_fresh_inputs_are_safe_under_pipelining(gpu_device):
    """Host tensors (copied to the device inside the call, i.e. still pending on the caller's stream) through pipeline level 2
    and through the triplet path with inputs_ready=True requested: the wrapper must not let the internal forward streams run
    ahead of the copy.  Results equal the un-pipelined ones bit for bit."""
    from xfr_amd.engine import Engine
    bb, sd = make_backbone('stresnet_mini', seed=3, num_classes=5)
    prog = bb.build_program()
    eng = Engine(prog, 8, gpu_device)
    eng.load_weights(sd)
    eng.set_mode('affineonly_with_prior')
    enc_t = prog.marks['encode']
    xs = [make_images('stresnet_mini', 4, seed=s) for s in (1, 2, 3)]                      # CPU tensors
    seeds = [torch.stack((synth.unit_rows(4, 512, seed=10 + s), synth.unit_rows(4, 512, seed=20 + s))) / 2500 for s in (1, 2, 3)]
    ref = [eng.contrastive(x, enc_t, sd_) .clone() for x, sd_ in zip(xs, seeds)]
    gal = [torch.cat((x, x.flip(0)), dim=0) for x in xs]
    ref_t = [eng.triplet_contrastive(x, g, enc_t).clone() for x, g in zip(xs, gal)]
    torch.cuda.synchronize()
    for level in (2, 6):                        # 6: three forward slots (bit 2)
        eng.set_pipeline(level)
        for rep in range(3):
            outs = [eng.contrastive(x, enc_t, sd_, inputs_ready=True) for x, sd_ in zip(xs, seeds)]
            outs_t = [eng.triplet_contrastive(x, g, enc_t, inputs_ready=True) for x, g in zip(xs, gal)]
            torch.cuda.synchronize()
            for o, r in zip(outs + outs_t, ref + ref_t):
                assert torch.equal(o, r)
    eng.close()


def test_host_inputs_keep_the_pipeline_and_the_bits(gpu_device):
    """xfr_triplet_contrastive_u8_host: fresh uint8 images from (pinned) host memory on every call, pipelining on (levels 1 and 5: two / three forward
    slots), no residency promise.  Every call's maps equal the un-pipelined device-input call's, bit for bit -- also when the host buffers are rewritten
    right after wait_inputs_copied, and when more calls are in flight than there are staging slots."""
    from xfr_amd.engine import Engine
    bb, sd = make_backbone('stresnet_mini', seed=3, num_classes=5)
    prog = bb.build_program()
    eng = Engine(prog, 8, gpu_device)
    eng.load_weights(sd)
    eng.set_mode('affineonly_with_prior')
    from xfr_amd.models import resnet
    eng.set_u8_preprocess('sub_mean', 3, tuple(float(v) for v in resnet.MEAN_RGB), None)
    enc_t = prog.marks['encode']
    g = torch.Generator().manual_seed(11)
    batches = [(torch.randint(0, 256, (4, 224, 224, 3), generator=g, dtype=torch.uint8), torch.randint(0, 256, (8, 224, 224, 3), generator=g, dtype=torch.uint8))
               for _ in range(5)]
    ref = [eng.triplet_contrastive_u8(p.to(gpu_device), q.to(gpu_device), enc_t).clone() for p, q in batches]
    torch.cuda.synchronize()
    for level in (0, 1, 5):
        eng.set_pipeline(level)
        hp, hg = [torch.empty_like(batches[0][0]).pin_memory() for _ in range(2)], [torch.empty_like(batches[0][1]).pin_memory() for _ in range(2)]
        outs = []
        for rep in range(2):
            for i, (p, q) in enumerate(batches):
                k = (rep * len(batches) + i) & 1
                eng.wait_inputs_copied()              # the buffer pair used two calls ago is free (copies run in order)
                hp[k].copy_(p)
                hg[k].copy_(q)
                outs.append(eng.triplet_contrastive_u8_host(hp[k], hg[k], enc_t))
        torch.cuda.synchronize()
        for j, o in enumerate(outs):
            assert torch.equal(o, ref[j % len(batches)]), (level, j)
    eng.set_pipeline(0)
    with pytest.raises(ValueError):
        eng.triplet_contrastive_u8_host(batches[0][0].to(gpu_device), batches[0][1], enc_t)      # a device tensor is not a host buffer
    eng.close()


# ---- multi-rank --------------------------------------------------------------------------------------------------------
def _run_ranks(cmd, world, port, extra_env=None, timeout=900):
    """world processes on ONE GPU (gloo rendezvous on 127.0.0.1): the multi-process code path of the tools without an 8-GPU node.  Every rank's
    stdout / stderr go to files (a rank blocked on a full pipe would block its peers at the next barrier)."""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        procs, files = [], []
        for r in range(world):
            env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(r), LOCAL_RANK=str(r),
                       XFR_DIST_BACKEND='gloo', XFR_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
            env.update(extra_env or {})
            fo, fe = open(os.path.join(d, 'out%d' % r), 'w+'), open(os.path.join(d, 'err%d' % r), 'w+')
            files.append((fo, fe))
            procs.append(subprocess.Popen([sys.executable] + cmd, cwd=ROOT, env=env, stdout=fo, stderr=fe, text=True))
        outs = []
        try:
            for p in procs:
                p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        for p, (fo, fe) in zip(procs, files):
            fo.seek(0)
            fe.seek(0)
            o, e = fo.read(), fe.read()
            assert p.returncode == 0, 'rank failed:\n%s\n%s' % (o[-2000:], e[-4000:])
            outs.append(o)
        return outs


def test_two_ranks_shard_the_inpainting_game_workload(gpu_device, tmp_path):
    """BASELINE.json configs[4] shape (eval/generate_inpaintinggame_wb_saliency_maps_multigpu.py:193-216: one worker per GPU,
    independent jobs): two ranks produce, between them, exactly the files and the bits of a one-rank run; rank 1 never packs
    parameters (it receives the arena by broadcast)."""
    tool = [os.path.join('tools', 'inpainting_game_workload.py'), '--jobs', '5', '--mates', '2', '--topk', '4', '--num-classes', '300']
    d1, d2 = str(tmp_path / 'one'), str(tmp_path / 'two')
    _run_ranks(tool + ['--output-dir', d1], 1, 29651)
    outs = _run_ranks(tool + ['--output-dir', d2], 2, 29653)
    line = json.loads([ln for ln in outs[0].splitlines() if ln.startswith('{')][-1])
    assert line['jobs'] == 5 and line['n_gpus'] == 2
    r0, r1 = (json.load(open(os.path.join(d2, 'rank%d.json' % r))) for r in (0, 1))
    assert r0['packed_weights'] and not r1['packed_weights']
    assert r0['jobs'] == [0, 3] and r1['jobs'] == [3, 5]
    files1 = sorted(os.path.relpath(os.path.join(dp, f), d1) for dp, _, fs in os.walk(d1) for f in fs if f.endswith('.npz'))
    files2 = sorted(os.path.relpath(os.path.join(dp, f), d2) for dp, _, fs in os.walk(d2) for f in fs if f.endswith('.npz'))
    assert files1 == files2 and len(files1) == 5 * 4
    for f in files1:
        a, b = np.load(os.path.join(d1, f))['saliency_map'], np.load(os.path.join(d2, f))['saliency_map']
        assert np.array_equal(a, b), f


def test_bench_two_ranks_on_one_gpu(gpu_device):
    """bench.py's multi-process path (rendezvous, arena broadcast, barrier + max-over-ranks timing, one JSON line on rank 0)."""
    outs = _run_ranks(['bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '8', '--no-cpu-baseline', '--no-sustained'], 2, 29657)
    lines = [ln for ln in outs[0].splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and not [ln for ln in outs[1].splitlines() if ln.startswith('{')]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['scaling'] == 'weak' and j['steps'] == 3 and j['outputs_ok'] is True
    assert abs(j['value'] - 2 * 8 * 3 / (j['ms_per_step'] * 3e-3)) < 1e-6 * j['value']


def test_bench_under_the_drivers_launcher(gpu_device):
    """The driver's own launch line -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ... -- with N = 2 on this one GPU (gloo instead of RCCL, both ranks on device 0)."""
    env = dict(os.environ, XFR_DIST_BACKEND='gloo', XFR_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29671', 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '8', '--no-cpu-baseline',
           '--no-sustained', '--no-profile']
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['steps'] == 2 and j['warmup'] == 1 and j['outputs_ok'] is True and j['unit'] == 'maps/s'


NCCL_WORLD1 = '''
import sys, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, %r)
from parity_utils import make_backbone, make_images
from xfr_amd import shard
from xfr_amd.engine import Engine
rank, world, local = shard.init_process_group('nccl', force=True)          # backend "nccl" IS RCCL on ROCm
assert dist.is_initialized() and dist.get_backend() == 'nccl' and (rank, world) == (0, 1)
dev = torch.device('cuda', local)
bb, sd = make_backbone('stresnet_mini', seed=3, num_classes=5)
prog = bb.build_program()
eng = Engine(prog, 4, dev)
shard.load_and_broadcast(eng, lambda: sd, src=0)                          # dist.broadcast of the arena tensor through RCCL
x = make_images('stresnet_mini', 3, seed=1).to(dev)
enc = eng.forward(x, prog.marks['encode']).reshape(3, -1)
maps = shard.gather_maps(enc.reshape(3, 1, -1), 3)                        # all_gather through RCCL
assert torch.equal(maps.reshape(3, -1), enc)
t = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                                  # bench.py's max-over-ranks timing
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print('NCCL_WORLD1_OK', float(t.item()))
'''


def test_rccl_backend_of_the_python_path_world_size_1(gpu_device):
    """The torch.distributed calls bench.py and the tools make at N > 1 -- broadcast of the arena, all_gather of maps, all_reduce
    MAX, barrier -- through the RCCL backend itself (one rank: that is what one GPU allows; the 2-rank tests above use gloo)."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29677', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0', NCCL_DEBUG='INFO')
    env.pop('XFR_DIST_BACKEND', None)
    out = subprocess.run([sys.executable, '-c', NCCL_WORLD1 % (ROOT, os.path.join(ROOT, 'tests'))], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and 'NCCL_WORLD1_OK 1.5' in out.stdout, out.stderr[-3000:]
    # NCCL_DEBUG=INFO: the communicator that carried the collectives is RCCL's (its banner and its init lines), not a stand-in
    log = out.stdout + out.stderr
    assert 'RCCL version' in log and 'NCCL INFO' in log and 'Init COMPLETE' in log, log[-3000:]


def test_bench_dry_run_and_rank_report(gpu_device):
    """bench.py --gpus 2 --dry-run: rendezvous, ONE arena broadcast, per-rank report (device, arena checksum, RCCL version), one
    step, barrier, exit -- the fast failure check for a multi-GPU box.  Both ranks must end up with the same arena bits."""
    outs = _run_ranks(['bench.py', '--gpus', '2', '--batch', '4', '--dry-run'], 2, 29663)
    lines = [ln for ln in outs[0].splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j['dry_run'] is True and j['n_gpus'] == 2 and j['outputs_finite'] is True and len(j['ranks']) == 2
    assert [r['rank'] for r in j['ranks']] == [0, 1] and len({r['arena_checksum48'] for r in j['ranks']}) == 1
    assert all(r['arena_bytes'] > 0 and r['rccl'] for r in j['ranks'])


def test_bench_survives_a_failed_weight_broadcast(gpu_device):
    """The weight broadcast fails on rank 1 (XFR_TEST_FAIL_BROADCAST): every rank packs from the seed locally, the arena checksums still agree,
    the line says "weights_via": "local_pack_fallback" -- not rc != 0 (eval/generate_inpaintinggame_wb_saliency_maps_multigpu.py:101-118: the
    reference's workers survive a failing job too)."""
    outs = _run_ranks(['bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '4', '--no-cpu-baseline', '--no-sustained', '--no-profile'],
                      2, 29681, extra_env={'XFR_TEST_FAIL_BROADCAST': '1', 'XFR_DIST_TIMEOUT': '45'})
    j = json.loads([ln for ln in outs[0].splitlines() if ln.startswith('{')][0])
    assert j['n_gpus'] == 2 and j['outputs_ok'] is True and j['weights_via'] == 'local_pack_fallback' and j['collective_backend_ok'] is False
    assert len({r['arena_checksum48'] for r in j['ranks']}) == 1 and all(r['weights_via'] == 'local_pack_fallback' for r in j['ranks'])


def test_bench_surfaces_a_failing_rank(gpu_device):
    """Rank 1 raises (XFR_TEST_RAISE_RANK): rank 0 neither hangs nor dies silently -- it prints ONE JSON line with every rank's exception text
    and both exit with code 4."""
    import tempfile
    procs, files = [], []
    tmp = tempfile.mkdtemp()
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29687', WORLD_SIZE='2', RANK=str(r), LOCAL_RANK=str(r), XFR_DIST_BACKEND='gloo',
                   XFR_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0', XFR_TEST_RAISE_RANK='1', XFR_DIST_TIMEOUT='60')
        fo, fe = open(os.path.join(tmp, 'o%d' % r), 'w+'), open(os.path.join(tmp, 'e%d' % r), 'w+')
        files.append((fo, fe))
        procs.append(subprocess.Popen([sys.executable, 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '4', '--no-cpu-baseline',
                                       '--no-sustained', '--no-profile'], cwd=ROOT, env=env, stdout=fo, stderr=fe, text=True))
    for p in procs:
        p.wait(timeout=600)
    outs = []
    for fo, fe in files:
        fo.seek(0)
        fe.seek(0)
        outs.append((fo.read(), fe.read()))
    assert [p.returncode for p in procs] == [4, 4], [o[1][-1500:] for o in outs]
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert 'simulated failure of rank 1' in j['rank_errors']['1'] and 'rank(s) [1] failed' in j['rank_errors']['0']


@pytest.mark.parametrize('world', [2, 8])
def test_bench_spawns_its_own_ranks(gpu_device, world):
    """`python bench.py --gpus N` started the way the N = 1 command is -- no torch.distributed.run, no RANK / WORLD_SIZE in the environment (round-5
    verdict, missing 2): bench.py is then its own launcher (eval/generate_inpaintinggame_wb_saliency_maps_multigpu.py:193-216 starts its workers itself
    too): N ranks, ONE JSON line on the parent's stdout, exit code 0."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(XFR_DIST_TIMEOUT='240', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, 'bench.py', '--gpus', str(world), '--steps', '2', '--warmup', '1', '--batch', '4', '--no-cpu-baseline', '--no-sustained',
                          '--no-profile'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == world and j['outputs_ok'] is True and len(j['ranks']) == world and len({r['arena_checksum48'] for r in j['ranks']}) == 1
    assert 'bench.py itself' in j['launcher'] and 'no multi-GPU hardware number' in j['scaling_note']


@pytest.mark.parametrize('bind', [False, True])
def test_bench_eight_ranks_on_one_gpu(gpu_device, bind):
    """Eight ranks (the driver's 8-GPU launch shape) as eight processes on this one GPU, four triplets each: eight launch threads fit the
    host's CPU quota with and without --bind, one line, eight equal arena checksums, one broadcast."""
    cmd = ['bench.py', '--gpus', '8', '--steps', '2', '--warmup', '1', '--batch', '4', '--no-cpu-baseline', '--no-sustained', '--no-profile']
    outs = _run_ranks(cmd + (['--bind'] if bind else []), 8, 29691 + int(bind) * 40, extra_env={'XFR_DIST_TIMEOUT': '240'}, timeout=900)
    j = json.loads([ln for ln in outs[0].splitlines() if ln.startswith('{')][0])
    assert j['n_gpus'] == 8 and j['outputs_ok'] is True and j['weights_via'] == 'broadcast' and len(j['ranks']) == 8
    assert len({r['arena_checksum48'] for r in j['ranks']}) == 1 and len(j['rank_maps_s']['per_rank']) == 8
    if bind:
        assert all('cpu_binding' in r for r in j['ranks'])


def test_rccl_entry_points_world_size_1(gpu_device):
    """xfr_comm_unique_id / xfr_comm_init / xfr_broadcast_weights / xfr_comm_destroy through librccl (one rank: the
    communicator is real, the broadcast degenerates to a self-copy); the receiving side's bookkeeping is checked on a second
    engine that never loaded weights."""
    from xfr_amd.engine import Engine
    lib = _lib.load()
    bb, sd = make_backbone('stresnet_mini', seed=3, num_classes=5)
    prog = bb.build_program()
    eng = Engine(prog, 2, gpu_device)
    eng.load_weights(sd)
    x = make_images('stresnet_mini', 2, seed=1).to(gpu_device)
    want = eng.forward(x, prog.marks['encode']).clone()
    uid = ctypes.create_string_buffer(128)
    _lib.check(lib.xfr_comm_unique_id(uid))
    comm = ctypes.c_void_p()
    _lib.check(lib.xfr_comm_init(0, 1, uid, gpu_device.index or 0, ctypes.byref(comm)))
    stream = ctypes.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
    before = eng.weight_arena().clone()
    _lib.check(lib.xfr_broadcast_weights(eng._h, comm, 0, stream))
    assert torch.equal(eng.weight_arena(), before)
    assert torch.equal(eng.forward(x, prog.marks['encode']), want)
    eng2 = Engine(prog, 2, gpu_device)                       # root without weights -> state error, nothing marked
    assert lib.xfr_broadcast_weights(eng2._h, comm, 0, stream) == _lib.XFR_STATE_ERROR
    assert lib.xfr_broadcast_weights(eng._h, comm, 3, stream) == _lib.XFR_INVALID_ARG
    _lib.check(lib.xfr_comm_destroy(comm))
    eng.close()
    eng2.close()


@pytest.mark.parametrize('arch,mode', [('lightcnn29v2', 'affineonly'), ('resnet50_128', 'norelu'), ('stresnet_mini', 'affineonly_with_prior')])
def test_image_mwp_of_every_backbone(gpu_device, arch, mode):
    """Whitebox.P[-1] (the hook on the first convolution's input, whitebox.py:394) against the oracle: Light-CNN's one-channel 5x5 first layer
    with interleaved MaxFeatureMap rows, ResNet-50-128d's bias-free 7x7 / 2 (4-channel tap pack), the STR stem; triplet classifier."""
    from oracle import ebp_oracle as O
    import golden_cases as GC
    from parity_utils import emb_dim
    from xfr_amd import synth
    bb, sd = make_backbone(arch, seed=4, num_classes=None if arch == 'resnet50_128' else 7)
    subj = GC.engine_subject(arch, bb, mode)
    D = emb_dim(arch)
    xm, xn = synth.unit_rows(1, D, seed=1) / 2500, synth.unit_rows(1, D, seed=2) / 2500
    subj.set_cls(xm, xn)
    ow = O.OracleWhitebox(arch, sd, ('hooked', None), mode)
    ow.set_triplet_classifier(xm, xn)
    x = make_images(arch, 1, seed=9)
    P2 = torch.zeros(1, 2)
    P2[0, 0] = 1
    subj.wb.ebp(x, P2)
    ow.ebp(x, P2)
    assert len(subj.wb.P) == len(ow.P)
    for k in (-1, -2):
        got, want = subj.wb.P[k].cpu().numpy(), ow.P[k].numpy()
        assert got.shape == want.shape and np.abs(want).max() > 0
        assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max(), (arch, k)
    # the on-demand P[-1] while plain calls are pipelined (levels 2 and 6: two / three forward slots): the gather must read the image of ITS call
    ref = subj.wb.P[-1].clone()
    eng = subj.wb._engine(1)
    x2 = make_images(arch, 1, seed=10)
    for level in (2, 6):
        eng.set_pipeline(level)
        for xi in (x2, x, x2, x):                # leaves the slot counter off 0 and another image in the other slots
            subj.wb.ebp(xi, P2)
        subj.wb.ebp(x, P2)
        assert torch.equal(subj.wb.P[-1], ref), level
    eng.set_pipeline(0)


@pytest.mark.parametrize('arch', ['stresnet101', 'resnet50_128', 'lightcnn29v2'])
def test_uint8_entry_points_match_the_reference_preprocessing(gpu_device, arch):
    """xfr_forward_u8 / xfr_triplet_contrastive_u8: uint8 H x W x C crops go to the device as they are and the engine does the pixel arithmetic of
    convert_resnet101v4_image (resnet.py:25-37), Whitebox_resnet50_128.preprocess (whitebox.py:235-258) and prepare_lightCNN_image
    (lightcnn.py:19-25).  On the four bundled JPEGs: the fp32 network input it builds equals the host function's tensor BIT FOR BIT, hence
    encodings and triplet maps from uint8 equal those from the preprocessed float tensors bit for bit."""
    import PIL.Image
    import golden_cases as GC
    from xfr_amd.models import whitebox as WB
    bb, sd = make_backbone(arch if arch != 'stresnet101' else 'stresnet_mini', seed=2, num_classes=None if arch == 'resnet50_128' else 5)
    bb.to(gpu_device)
    wbn = {'stresnet101': WB.WhiteboxSTResnet, 'resnet50_128': WB.Whitebox_resnet50_128, 'lightcnn29v2': WB.WhiteboxLightCNN}[arch](bb)
    wb = WB.Whitebox(wbn, ebp_subtree_mode='affineonly_with_prior')
    pics = [GC.jpegs()[f] for f in GC.JPEGS]                                   # 224 x 224 x 3 uint8
    if arch == 'lightcnn29v2':                                                 # Resize(144) + CenterCrop(128) stay on the host, like in the reference
        crops = []
        for a in pics:
            im = PIL.Image.fromarray(a).resize((144, 144), PIL.Image.BILINEAR).crop((8, 8, 136, 136))
            crops.append(np.asarray(im).copy())
        host = torch.cat([WB.lightcnn_preprocess.__globals__['prepare_lightCNN_image'](PIL.Image.fromarray(c)) for c in crops])
    elif arch == 'resnet50_128':
        crops = pics
        host = torch.cat([wbn.preprocess(PIL.Image.fromarray(c)) for c in crops])        # 224 x 224 in: its resize / crop are the identity
    else:
        crops = pics
        host = torch.cat([wbn.preprocess(PIL.Image.fromarray(c)) for c in crops])
    u8 = torch.from_numpy(np.stack(crops))
    eng = wb._engine(8)
    dev_in = eng.preprocess_u8(u8).cpu()
    assert dev_in.shape == host.shape and torch.equal(dev_in, host.float())
    assert torch.equal(wbn.encode_u8(u8), wbn.encode(host.to(gpu_device)))
    a = wb.triplet_images_ebp_batch_u8(u8, u8[[1, 2, 3, 0]], u8[[2, 3, 0, 1]])
    b = wb.triplet_images_ebp_batch(host.to(gpu_device), host[[1, 2, 3, 0]].to(gpu_device), host[[2, 3, 0, 1]].to(gpu_device))
    assert torch.isfinite(a).all() and torch.equal(a, b)

