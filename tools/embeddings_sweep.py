#!/usr/bin/env python
"""the forward-only `embeddings` sweep of the blackbox / inpainting-game code (python/xfr/models/
blackbox.py:366-414 scores ~6500 RISE-masked copies of one probe; whitebox.py:747-785 is the batched encode it calls) on
synthetic data: N masked probes, encoded in batches that stay in HBM.  Prints one JSON line (images/s).
    python tools/embeddings_sweep.py --masks 6500 --batch 128
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--masks', type=int, default=6500)
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--cells', type=int, default=7, help='RISE grid: cells x cells random mask, bilinearly upsampled')
    args = ap.parse_args()
    import torch
    from xfr_amd import synth
    from xfr_amd.models import resnet, whitebox as WB

    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    bb = resnet.ResNet([3, 4, 23, 3], num_classes=2)
    bb.load_state_dict(synth.synth_state_dict(bb, seed=0))
    bb.to(dev)
    wbn = WB.WhiteboxSTResnet(bb)
    wbn.default_max_batch = args.batch
    wb = WB.Whitebox(wbn)
    wb.batch_size = args.batch
    probe = synth.synth_smooth_images(1, (3, 224, 224), seed=7, mean=resnet.MEAN_RGB).to(dev)
    g = torch.Generator(device='cpu').manual_seed(0)

    def masked_batch(n):
        grid = (torch.rand((n, 1, args.cells, args.cells), generator=g) < 0.5).float().to(dev)
        m = torch.nn.functional.interpolate(grid, size=(224, 224), mode='bilinear', align_corners=False)
        return (probe * m).contiguous()

    ref = wb.encode(probe)
    wb.encode(masked_batch(args.batch))                     # warm-up: engine at this batch size
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done, sims = 0, []
    while done < args.masks:
        n = min(args.batch, args.masks - done)
        emb = wb.encode(masked_batch(n))
        sims.append(torch.nn.functional.cosine_similarity(emb, ref.expand_as(emb)))   # the score RISE accumulates
        done += n
    sims = torch.cat(sims)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({'workload': 'RISE-style embeddings sweep, ResNet-101 224x224, synthetic', 'masks': args.masks, 'batch': args.batch,
                      'seconds': dt, 'images_per_s': args.masks / dt, 'forward_TFLOP_per_s': args.masks * 14.419e9 / dt / 1e12,
                      'mean_similarity': float(sims.mean())}))


if __name__ == '__main__':
    main()
