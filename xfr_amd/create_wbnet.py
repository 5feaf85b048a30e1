"""Mirror of eval/create_wbnet.py: net name -> Whitebox with the reference's per-network defaults.

Same names, default subtree modes, match thresholds and Platt scalings as the reference (eval/create_wbnet.py:10-137).
The reference resolves fixed checkpoint paths under its repository (all of them git-LFS pointers in the public tree);
here the checkpoint is an explicit argument, and when it is None the backbone keeps its seeded random initialisation
(a warning is emitted) so that the factory is usable for benchmarks and tests without the weights.
"""
import warnings

import torch

from .models import lightcnn, resnet, resnet50_128, whitebox

NET_DEFAULTS = {
    # name: (default ebp_subtree_mode, match_threshold, platts_scaling)      eval/create_wbnet.py
    'resnetv6_pytorch': ('norelu', 0.9636, 15.05),                   # :24-46
    'resnetv4_pytorch': ('norelu', 0.9722, 16.61),                   # :48-72
    'vggface2_resnet50': ('norelu', 0.896200, 15.921608),            # :74-100
    'lightcnn': ('affineonly_with_prior', 0.829200, 10.877741),      # :102-132
}


def create_wbnet(net_name, device=None, ebp_version=None, ebp_subtree_mode=None, weights_path=None):
    """eval/create_wbnet.py:10-137."""
    if device is None:
        device = torch.device('cuda:0' if torch.cuda.is_available() else 'cpu')     # :17-20
    if ebp_version is not None and ebp_version < 4:
        raise DeprecationWarning('EBP version must be >= 4')                        # :22-23
    if net_name not in NET_DEFAULTS:
        raise NotImplementedError('create_wbnet does not implemented network "%s"' % net_name)   # :133-137
    default_mode, thr, platt = NET_DEFAULTS[net_name]
    if ebp_subtree_mode is None:
        ebp_subtree_mode = default_mode
    if weights_path is None:
        warnings.warn('create_wbnet(%s): no checkpoint given, the backbone keeps its random initialisation' % net_name)
    if net_name in ('resnetv6_pytorch', 'resnetv4_pytorch'):
        model = resnet.resnet101v6(weights_path, device)
        model.to(device)
        wbnet = whitebox.WhiteboxSTResnet(model)
    elif net_name == 'vggface2_resnet50':
        if ebp_version is not None:
            warnings.warn('ebp_version %s is ignored for %s' % (ebp_version, net_name))   # :78-82
        model = resnet50_128.resnet50_128(weights_path)
        model.to(device)
        wbnet = whitebox.Whitebox_resnet50_128(model)
    else:
        model = lightcnn.LightCNN_29Layers_v2(num_classes=80013)
        if weights_path is not None:
            model.load_state_dict(lightcnn.Load_Checkpoint(weights_path))
        model.to(device)
        wbnet = whitebox.WhiteboxLightCNN(model)
    wb = whitebox.Whitebox(wbnet, ebp_subtree_mode=ebp_subtree_mode, ebp_version=ebp_version).to(device)
    wb.match_threshold = thr
    wb.platts_scaling = platt
    return wb
