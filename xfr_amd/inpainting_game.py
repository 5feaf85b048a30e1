"""Callers of the whitebox path in the reference's inpainting-game generator, with the reference's names and signatures
(python/xfr/inpainting_game/generate_whitebox_saliency.py:79-214), so that its job loop
(eval/generate_inpaintinggame_wb_saliency_maps_multigpu.py:200-215) can import them from here unchanged:

    mean_ebp(wb, probe_im, net_name, ebp_version, device)                                            :207-214
    run_contrastive_triplet_ebp(wb, im_mates, im_nonmates, probe_im, net_name, ebp_version, truncate_percent, device)   :79-115
    run_weighted_subtree_triplet_ebp(wb, im_mates, im_nonmates, probe_im, net_name, subtree_mode_weighted, ebp_version,
                                     device, topk=1)                                                 :119-205

Images are what `xfr.utils.image_loader` yields (H x W x 3 float in [0, 1]) or uint8 arrays; `wb` is an
xfr_amd.models.whitebox.Whitebox.  The drop-in functions encode the k mate and k non-mate images one forward at a time like
the reference (ENCODE_ONE_BY_ONE below), so they reproduce its arithmetic; run_jobs_batched is the fast, additive path that
batches the encodes.  `net_name` is accepted for signature compatibility and unused, as in the reference.
"""
import torch

# ebp_version -> (do_max_subtree, do_mated_similarity_gating) of weighted_subtree_ebp: generate_whitebox_saliency.py:171-194;
# any other version keeps the function's defaults (:142-143)
SUBTREE_VERSIONS = {7: (True, True), 8: (False, True), 9: (True, False), 10: (True, True), 11: (True, True), 12: (False, True)}


# The reference encodes the k images one forward at a time (:85-92).  Batching them changes the fp32 summation order of the
# convolutions, i.e. the encodings in their last bits (2e-8 measured on the CPU path) -- harmless, except that with nearly
# parallel mate / non-mate directions contrastive EBP amplifies classifier perturbations by ~5e4 (tests/test_c2.py).  The
# drop-in callers therefore default to the reference's forward-by-forward arithmetic; set to False to encode each gallery as one
# batch (run_jobs_batched always batches).
ENCODE_ONE_BY_ONE = True


def mean_encoding(wb, images, device):
    """Unit-normalised mean of the encodings of `images` (:87-98): 1 x D tensor on `device`."""
    x = torch.cat([wb.convert_from_numpy(im) for im in images], dim=0).to(device)
    if ENCODE_ONE_BY_ONE:
        enc = torch.cat([wb.encode(x[i:i + 1]).detach() for i in range(x.shape[0])], dim=0)
    else:
        enc = wb.encode(x).detach()
    avg = torch.mean(enc.reshape(enc.shape[0], 1, -1), axis=0)
    return avg / torch.norm(avg)


def mean_ebp(wb, probe_im, net_name, ebp_version, device):
    """EBP saliency at the first convolution for a uniform prior over all classes of the hooked classifier."""
    x_probe = wb.convert_from_numpy(probe_im).to(device)
    P = torch.ones((1, wb.net.num_classes())).to(device)
    return wb.ebp(x_probe, P)


def run_contrastive_triplet_ebp(wb, im_mates, im_nonmates, probe_im, net_name, ebp_version, truncate_percent, device):
    """Contrastive (truncate_percent None) or truncated contrastive EBP of the probe against the averaged mates / non-mates."""
    avg_x_mate = mean_encoding(wb, im_mates, device)
    avg_x_nonmate = mean_encoding(wb, im_nonmates, device)
    img_probe = wb.convert_from_numpy(probe_im).to(device)
    wb.net.set_triplet_classifier((1.0 / 2500.0) * avg_x_mate, (1.0 / 2500.0) * avg_x_nonmate)
    if truncate_percent is None:
        return wb.contrastive_ebp(img_probe, k_poschannel=0, k_negchannel=1)
    return wb.truncated_contrastive_ebp(img_probe, k_poschannel=0, k_negchannel=1, percentile=truncate_percent)


def run_weighted_subtree_triplet_ebp(wb, im_mates, im_nonmates, probe_im, net_name, subtree_mode_weighted, ebp_version, device,
                                     topk=1):
    """Weighted subtree EBP with the parameterisation `ebp_version` selects; the classifier rows are the unit-norm averages
    themselves here (no 1/2500 factor, :134)."""
    avg_x_mate = mean_encoding(wb, im_mates, device)
    avg_x_nonmate = mean_encoding(wb, im_nonmates, device)
    img_probe = wb.convert_from_numpy(probe_im).to(device)
    wb.net.set_triplet_classifier(avg_x_mate, avg_x_nonmate)
    do_max_subtree, do_mated_similarity_gating = SUBTREE_VERSIONS.get(ebp_version, (False, False))
    img_subtree, _, _, _ = wb.weighted_subtree_ebp(img_probe, k_poschannel=0, k_negchannel=1, topk=topk, verbose=False,
                                                   do_max_subtree=do_max_subtree, subtree_mode=subtree_mode_weighted,
                                                   do_mated_similarity_gating=do_mated_similarity_gating)
    return img_subtree


def run_jobs_batched(wb, jobs, net_name, subtree_mode_weighted, ebp_version, device, topk=32, methods=None, timings=None):
    """Additive: the four saliency methods of generate_wb_smaps (:295-399) for a GROUP of independent jobs in shared launches.
    jobs: list of (im_mates, im_nonmates, probe_im).  Per job the calls of mean_ebp (always over the HOOKED N-way classifier,
    which is what the generator's job loop has installed at that point; the drop-in mean_ebp uses whatever classifier is
    current), run_contrastive_triplet_ebp (truncate_percent None and 20) and run_weighted_subtree_triplet_ebp above, with the
    galleries encoded as batches (last-bit differences in the encodings, see ENCODE_ONE_BY_ONE); here all probes form one batch, all mate /
    non-mate images one encode batch, and each method one (or a few) engine calls: Whitebox.ebp at N probes,
    contrastive_triplet_ebp_batch, weighted_subtree_ebp_batch.  Returns {method: [map per job]} with the method keys
    'meanEBP', 'contrastive', 'truncated', 'weighted-subtree'.  Needs the hooked classifier for meanEBP (restored afterwards)
    and ebp_version 6 maps (float32) for the batched tails.  The three triplet methods run the same probes through the
    network: they share one forward pass (xfr_engine_hold_forward).  timings: optional dict that receives seconds per phase
    (synchronises the device between phases)."""
    import time
    methods = methods or ('meanEBP', 'contrastive', 'truncated', 'weighted-subtree')
    n = len(jobs)
    clock = [time.perf_counter()]

    def lap(name):
        if timings is not None:
            torch.cuda.synchronize()
            t = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + t - clock[0]
            clock[0] = t
    probes = torch.cat([wb.convert_from_numpy(j[2]) for j in jobs], dim=0).to(device)
    out = {}
    saved = wb.net._classifier
    if 'meanEBP' in methods:
        wb.net._classifier = None       # the hooked N-way classifier (mean_ebp above uses whatever classifier is current)
        try:
            m = wb.ebp(probes, torch.ones((1, wb.net.num_classes())))
        finally:
            wb.net._classifier = saved
        out['meanEBP'] = list(m.reshape((n,) + m.shape[-2:]))
        lap('meanEBP')
    if len(set(methods) - {'meanEBP'}) == 0:
        return out
    km = [len(j[0]) for j in jobs]
    kn = [len(j[1]) for j in jobs]
    gallery = torch.cat([wb.convert_from_numpy(im) for j in jobs for im in list(j[0]) + list(j[1])], dim=0).to(device)
    enc = torch.cat([wb.encode(gallery[i:i + wb.batch_size]).detach() for i in range(0, gallery.shape[0], wb.batch_size)], dim=0)
    xm, xn, o = [], [], 0
    for a, b in zip(km, kn):
        m_ = enc[o:o + a].mean(dim=0, keepdim=True)
        n_ = enc[o + a:o + a + b].mean(dim=0, keepdim=True)
        xm.append(m_ / torch.norm(m_))
        xn.append(n_ / torch.norm(n_))
        o += a + b
    xm, xn = torch.cat(xm, dim=0), torch.cat(xn, dim=0)
    lap('encodes')
    eng = wb._engine(n)
    probes, _ = eng._prep(probes)
    eng.hold_forward(True)
    try:
        if 'contrastive' in methods:
            out['contrastive'] = list(wb.contrastive_triplet_ebp_batch(probes, xm / 2500.0, xn / 2500.0).cpu().numpy())
            lap('contrastive')
        if 'truncated' in methods:
            out['truncated'] = list(wb.contrastive_triplet_ebp_batch(probes, xm / 2500.0, xn / 2500.0, percentile=20).cpu().numpy())
            lap('truncated')
        if 'weighted-subtree' in methods:
            do_max_subtree, gating = SUBTREE_VERSIONS.get(ebp_version, (False, False))
            res = wb.weighted_subtree_ebp_batch(probes, xm, xn, k_poschannel=0, topk=topk, do_max_subtree=do_max_subtree,
                                                do_mated_similarity_gating=gating, subtree_mode=subtree_mode_weighted)
            out['weighted-subtree'] = [r[0] for r in res]
            lap('weighted-subtree')
    finally:
        eng.hold_forward(False)
    return out


def shorten_subtree_mode(ebp_subtree_mode):
    """:216-219"""
    return 'awp' if ebp_subtree_mode == 'affineonly_with_prior' else ebp_subtree_mode


__all__ = ['mean_ebp', 'run_contrastive_triplet_ebp', 'run_weighted_subtree_triplet_ebp', 'run_jobs_batched', 'shorten_subtree_mode', 'mean_encoding',
           'SUBTREE_VERSIONS']
