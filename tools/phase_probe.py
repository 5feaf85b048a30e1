#!/usr/bin/env python
"""phase_probe.py -- where a GEMM workgroup's life goes WHILE the timed three-stream step runs (not in isolation).

Every conv_gemm workgroup stamps entry / first operands / end of K loop / end of exchange / exit (xfr_debug_conv_stamps) and the
shape of the launch it belongs to.  In sampled mode every launch leaves up to 256 evenly spaced workgroup records in a region of its own.  Per launch shape (kernel, K, Cout, M): how long the prologue, the K loop and the epilogue took, and the
K loop's stretch = its duration / the time its MFMAs need alone on a SIMD (K x 32 cycles per wave at the measured clock).
With the launch log (tiles per launch) the sample is weighted into chip-level figures: workgroups resident per CU, waves inside
their K loop per SIMD, and the MFMA duty those imply.

    python tools/phase_probe.py [--steps 12] [--serial]
"""
import argparse
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--serial', action='store_true')
    args = ap.parse_args()
    import numpy as np
    import torch
    from xfr_amd import _lib, synth, tuning
    from xfr_amd.engine import Engine
    from xfr_amd.models import resnet
    lib = _lib.load()
    dev = torch.device('cuda', 0)
    B = args.batch
    bb = resnet.ResNet([3, 4, 23, 3], num_classes=2)
    prog = bb.build_program()
    eng = Engine(prog, 2 * B, dev)
    eng.load_weights(synth.synth_state_dict(bb, seed=0, recipe='mild'))
    eng.set_mode('affineonly_with_prior')
    if not args.serial:
        eng.set_pipeline(True)
    imgs = synth.bench_images(B, (3, 224, 224), seed=1234, mean=resnet.MEAN_RGB).to(dev)
    gallery, probes = imgs[:2 * B].contiguous(), imgs[2 * B:3 * B].contiguous()
    enc_t = prog.marks['encode']
    if args.serial:
        eng.set_profile(True)

    def step():
        return eng.triplet_contrastive(probes, gallery, enc_t, 1.0 / 2500.0, None, inputs_ready=not args.serial)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    # 1. the launch log: tiles per launch shape and the step time
    csv = tuning.record_launch_log(step, 4, dev)
    rows = [l.strip().split(',') for l in open(csv)][1:]
    os.unlink(csv)
    per_step = len(rows) // 6
    shape_n = collections.Counter()          # (K, Cout, M) -> launches per step
    for r in rows[per_step:2 * per_step]:
        shape_n[(int(r[4]) & 0xffff, int(r[2]) * 1, int(r[5]))] += 1
    ends = [max(int(r[10]) for r in rows[j * per_step:(j + 1) * per_step]) for j in range(6)]
    step_ticks = (ends[4] - ends[0]) / 4.0
    # 2. the stamps, sampled mode: one region of 256 records per launch of a step (the last step that ran leaves its records)
    regions = per_step + 17
    nwg = regions * 256
    st = torch.zeros((nwg * 32,), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    _lib.check(lib.xfr_debug_conv_stamps(st.data_ptr(), -nwg))
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    _lib.check(lib.xfr_debug_conv_stamps(None, 0))
    v = st.cpu().numpy().reshape(nwg, 4, 8).astype(np.uint64)
    meta = v[:, 0, 6]
    K = ((meta >> np.uint64(8)) & np.uint64(0xffff)).astype(np.int64)
    M = ((meta >> np.uint64(24)) & np.uint64(0x3fffff)).astype(np.int64)
    Co = ((meta >> np.uint64(46)) & np.uint64(0x1fff)).astype(np.int64)
    kind = ((meta >> np.uint64(60)) & np.uint64(1)).astype(np.int64)
    t = v[:, :, :5].astype(np.float64)
    cyc = v[:, 0, 7].astype(np.float64)
    whole = (t[:, :, 0].min(axis=1) > 0) & (t[:, :, 4].min(axis=1) > 0) & (t[:, :, 4].max(axis=1) > t[:, :, 0].min(axis=1))
    # a record mixed from two launches (another stream's workgroup with the same index overwrote part of it) is not monotone
    mono = np.all(np.diff(t[:, 0, :], axis=1) >= 0, axis=1)
    ok = whole & mono
    life = (t[:, :, 4].max(axis=1) - t[:, :, 0].min(axis=1))
    ok &= (life < 1e5)
    ghz = np.median(cyc[ok] / (t[ok, 0, 4] - t[ok, 0, 0]) / 10.0)
    print('# %d consistent workgroup records, effective clock %.3f GHz, step %.3f ms, %d GEMM launches per step' % (ok.sum(), ghz, step_ticks * 1e-5, per_step))
    groups = collections.defaultdict(list)
    for i in np.nonzero(ok)[0]:
        groups[(int(kind[i]), int(K[i]), int(Co[i]), int(M[i]))].append(i)
    print('# kernel K Cout M | launches/step tiles/launch | records | prologue / K loop / exchange+epilogue us (median) | life | K-loop stretch (x its MFMA-alone time)')
    tot_life = tot_k = tot_mfma = 0.0
    covered = 0
    table = []
    slots = {}
    for key, ix in groups.items():
        kd, k, co, m = key
        ix = np.array(ix)
        pro = np.median((t[ix, :, 1] - t[ix, :, 0]).max(axis=1)) * 1e-2
        kl = np.median((t[ix, :, 2] - t[ix, :, 1]).max(axis=1)) * 1e-2
        ep = np.median((t[ix, :, 4] - t[ix, :, 2]).max(axis=1)) * 1e-2
        lf = np.median(life[ix]) * 1e-2
        kpad = (k + 31) // 32 * 32
        alone = kpad * 32.0 / (ghz * 1e3)                     # us: K/2 MFMAs of 64 cycles per wave (both kernels)
        n = shape_n.get((k, co, m), 0)
        tiles = ((m + 63) // 64) * ((co + 63) // 64)
        kls = (t[ix, :, 2] - t[ix, :, 1]).max(axis=1) * 1e-2 / alone
        wid = (v[ix, 0, 5] & np.uint64(15)).astype(np.int64)          # HW_ID.wave_id: the slot wave 0 got on its SIMD (lowest free one)
        slots[key] = np.bincount(wid, minlength=10)[:10] / float(len(ix))
        table.append((n * tiles * lf, kd, k, co, m, n, tiles, len(ix), pro, kl, ep, lf, kl / alone) + tuple(np.percentile(kls, [10, 25, 75, 90])))
        if n:
            covered += n
            tot_life += n * tiles * lf
            tot_k += n * tiles * kl
            tot_mfma += n * tiles * alone
    for row in sorted(table, reverse=True)[:24]:
        print('%s K %5d Cout %4d M %7d | %3d x %5d | %5d | %5.1f / %6.1f / %5.1f | %6.1f | %.2f (p10/25/75/90 %.2f %.2f %.2f %.2f)' % ((('ks' if row[1] else 'k1'),) + row[2:]))
    print('# SIMD slot (HW_ID.wave_id) a workgroup\'s wave 0 started in: share of the records per slot 0..9')
    for row in sorted(table, reverse=True)[:12]:
        print('%s K %5d Cout %4d M %7d | %s' % (('ks' if row[1] else 'k1'), row[2], row[3], row[4], ' '.join('%.2f' % x for x in slots[(row[1], row[2], row[3], row[4])])))
    T = step_ticks * 1e-2                                     # us
    print('# launches per step covered by the sample: %d of %d' % (covered, per_step))
    print(json.dumps({'step_ms': T * 1e-3, 'clock_GHz': ghz, 'workgroups_resident_per_CU': tot_life / (T * 256),
                      'waves_in_K_loop_per_SIMD': tot_k / (T * 256), 'mfma_duty_implied': tot_mfma / (T * 256),
                      'covered_launches': covered, 'launches_per_step': per_step}))


if __name__ == '__main__':
    main()
