"""Callers of the whitebox path in the reference's inpainting-game generator, with the reference's names and signatures
(python/xfr/inpainting_game/generate_whitebox_saliency.py:79-214), so that its job loop
(eval/generate_inpaintinggame_wb_saliency_maps_multigpu.py:200-215) can import them from here unchanged:

    mean_ebp(wb, probe_im, net_name, ebp_version, device)                                            :207-214
    run_contrastive_triplet_ebp(wb, im_mates, im_nonmates, probe_im, net_name, ebp_version, truncate_percent, device)   :79-115
    run_weighted_subtree_triplet_ebp(wb, im_mates, im_nonmates, probe_im, net_name, subtree_mode_weighted, ebp_version,
                                     device, topk=1)                                                 :119-205

Images are what `xfr.utils.image_loader` yields (H x W x 3 float in [0, 1]) or uint8 arrays; `wb` is an
xfr_amd.models.whitebox.Whitebox.  Difference in execution, not in results: the k mate and k non-mate images are encoded as
one batch instead of one forward each.  `net_name` is accepted for signature compatibility and unused, as in the reference.
"""
import torch

# ebp_version -> (do_max_subtree, do_mated_similarity_gating) of weighted_subtree_ebp: generate_whitebox_saliency.py:171-194;
# any other version keeps the function's defaults (:142-143)
SUBTREE_VERSIONS = {7: (True, True), 8: (False, True), 9: (True, False), 10: (True, True), 11: (True, True), 12: (False, True)}


# The reference encodes the k images one forward at a time (:85-92).  Batching them changes the fp32 summation order of the
# convolutions, i.e. the encodings in their last bits (2e-8 measured on the CPU path) -- harmless, except that with nearly
# parallel mate / non-mate directions contrastive EBP amplifies classifier perturbations by ~5e4 (tests/test_c2.py).  Set to
# True to reproduce the reference's forward-by-forward arithmetic exactly.
ENCODE_ONE_BY_ONE = False


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


def shorten_subtree_mode(ebp_subtree_mode):
    """:216-219"""
    return 'awp' if ebp_subtree_mode == 'affineonly_with_prior' else ebp_subtree_mode


__all__ = ['mean_ebp', 'run_contrastive_triplet_ebp', 'run_weighted_subtree_triplet_ebp', 'shorten_subtree_mode', 'mean_encoding',
           'SUBTREE_VERSIONS']
