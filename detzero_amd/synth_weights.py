"""Seeded synthetic detector weights (there are no checkpoints offline): what every parity test, `smoke()` and `bench.py` run on.

Fixtures, not the detection path: nothing in detzero_amd/ calls into this module, and the one place that evaluates layers with torch's
own functional ops (`uniform_field_response`: the far-field value of the heat map on a 4 x 4 torus, on the CPU, to place a bias) computes
a property of a weight set - no frame goes through it.  `detzero_amd.centerpoint.synth_detector` resolves to the function here."""
import math

import torch
import torch.nn as nn

from .det_modules import SparseConv3d, SubMConv3d
from .lib import DetZeroHipError

# mean number of ACTIVE taps per output site of each sparse convolution on a 64-beam sweep at 0.1 m voxels (rulebook
# pairs / output sites of the seed-0 frame): the fan-in a variance-preserving initialisation has to count - not the 27 of the kernel
_SWEEP_TAPS = {'conv_input': 3.6, 'conv1': 3.6, 'conv2.0': 2.2, 'conv2': 7.9, 'conv3.0': 4.6, 'conv3': 10.5, 'conv4.0': 6.2, 'conv4': 12.9,
               'conv_out': 2.1}


@torch.no_grad()
def uniform_field_response(backbone2d, dense_head, head_name='hm'):
    """Per-class value of a CenterHead output far from any data and from the border: the response of the dense stage to an all-zero BEV
    map, which is periodic with the upsampling stride - computed exactly on a 4 x 4 torus (circular padding), on the CPU, from the
    modules' own weights.  A property of a weight set (used by `variance_preserving_init` to place the heat-map bias), not a detection
    path: nothing is detected here.  Returns one (classes,) tensor per head (the largest phase of the period)."""
    import torch.nn.functional as F

    def run(seq, x):
        for m in seq:
            if isinstance(m, nn.Sequential):
                x = run(m, x)
            elif isinstance(m, nn.ZeroPad2d):
                x = F.pad(x, (1, 1, 1, 1), mode='circular')
            elif isinstance(m, nn.ConvTranspose2d):
                x = F.conv_transpose2d(x, m.weight, m.bias, stride=m.stride)
            elif isinstance(m, nn.Conv2d):
                if m.padding[0]:
                    x = F.pad(x, (m.padding[1],) * 2 + (m.padding[0],) * 2, mode='circular')
                x = F.conv2d(x, m.weight, m.bias, stride=m.stride)
            elif isinstance(m, nn.BatchNorm2d):
                x = F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, False, 0.0, m.eps)
            elif isinstance(m, nn.ReLU):
                x = torch.relu(x)
            else:
                raise DetZeroHipError('uniform_field_response: unexpected layer %s' % type(m).__name__)
        return x
    x = torch.zeros(1, backbone2d.input_channels, 4, 4)
    ups = []
    for blk, de in zip(backbone2d.blocks, backbone2d.deblocks):
        x = run(blk, x)
        ups.append(run(de, x))
    x = run(dense_head.shared_conv, torch.cat(ups, dim=1))
    return [run(getattr(h, head_name), x).amax((0, 2, 3)) for h in dense_head.heads_list]


@torch.no_grad()
def variance_preserving_init(model, seed=0, ring=0.15, branch=0.5, down=0.4, up2=0.5, hm_gain=3.0, hm_floor=-4.0):
    """Second synthetic weight set (`synth_detector(gain='preserve')`): a detector whose boxes DEPEND ON THE FRAME.  The default
    initialisers shrink the data-dependent signal ~6x in variance per layer (kaiming_uniform(a=sqrt 5) is sqrt 6 short of He, and a sparse
    kernel sees 2-13 of its 27 taps), so after the 25 dense layers only the border response of the zero padding is left and every frame
    yields the same boxes (round-5 review).  Here:
      * every hidden convolution is re-drawn N(0, 2 / fan_in) with the fan-in that is really summed: active taps x Cin for the sparse
        kernels (`_SWEEP_TAPS`), Cin x (1 + 8 ring^2) for the 3 x 3 BEV / head kernels, whose eight outer taps carry `ring` times the
        centre tap's amplitude (a centre-heavy kernel keeps the response on the cell the data is in; trained BEV kernels are centre-heavy
        too), Cin for the deblocks (the 2 x 2 upsampling one, which spreads a coarse cell over four fine ones, at `up2` of that);
        the first layer's input columns are scaled to O(1) (x, y in units of 40 m);
      * the second convolution of a residual block carries `branch`, a strided convolution `down`, so that the sparse stages stay O(1);
      * the output layers keep their default draw, scaled so that the decode is as well-conditioned as a trained head's: centre offsets
        0.5 +- 0.2 of a cell, log-sizes 0.25 around a car's (exp() of a log-size of 4 would turn a 1e-4 error into 5e-3 m), the (cos, sin)
        pair 0.3 around a unit vector (atan2 of a pair of norm 0.05 turns 1e-4 into 2e-3 rad; a trained head's pair has norm ~ 1), heat map
        x `hm_gain` with its bias placed `hm_floor` below zero for a cell far from any data (`uniform_field_response`) - sigmoid(-4) = 0.018 is under
        SCORE_THRESH, so a box exists only where the frame's points raised the heat map.
    BatchNorm statistics / affine terms are left to the caller (synth_detector randomises them BEFORE calling this)."""
    g = torch.Generator().manual_seed(1000 + seed)
    for name, m in model.named_modules():
        if isinstance(m, (SubMConv3d, SparseConv3d)) and name.startswith('backbone3d.'):
            parts = name[len('backbone3d.'):].split('.')
            key = parts[0] + ('.0' if parts[0] in ('conv2', 'conv3', 'conv4') and parts[1] == '0' else '')
            cin = m.weight.shape[-1]
            std = math.sqrt(2.0 / (_SWEEP_TAPS[key] * cin))
            if parts[-1] == 'conv2':
                std *= branch
            if not m.subm:
                std *= down
            m.weight.copy_(torch.randn(m.weight.shape, generator=g) * std)
            if key == 'conv_input':
                col = torch.ones(cin)
                col[:3] = torch.tensor([1 / 40.0, 1 / 40.0, 0.5])
                m.weight.mul_(col)
        elif isinstance(m, nn.ConvTranspose2d) and name.startswith('backbone2d.'):
            std = math.sqrt(2.0 / m.weight.shape[0]) * (up2 if m.stride[0] > 1 else 1.0)
            m.weight.copy_(torch.randn(m.weight.shape, generator=g) * std)
        elif isinstance(m, nn.Conv2d) and (name.startswith('backbone2d.') or name.startswith('dense_head.')):
            if name.startswith('dense_head.heads_list') and name.split('.')[-2] != '0':
                continue                                     # output layers: below
            taps = torch.full(m.weight.shape[2:], float(ring))
            taps[m.weight.shape[2] // 2, m.weight.shape[3] // 2] = 1.0
            std = math.sqrt(2.0 / (m.weight.shape[1] * float(taps.pow(2).sum())))
            m.weight.copy_(torch.randn(m.weight.shape, generator=g) * std * taps)
    for hl in model.dense_head.heads_list:
        for head, gain, bias in (('center', 0.15, [0.5, 0.5]), ('center_z', 0.5, [1.0]), ('dim', 0.12, [1.2, 0.6, 0.4]), ('rot', 0.3, [0.8, 0.6]),
                                 ('iou', 0.5, [0.6])):
            if hasattr(hl, head):
                out = getattr(hl, head)[1]
                out.weight.mul_(gain)
                out.bias.copy_(torch.tensor(bias))
        hl.hm[1].weight.mul_(hm_gain)
        hl.hm[1].bias.zero_()
    for hl, z0 in zip(model.dense_head.heads_list, uniform_field_response(model.backbone2d, model.dense_head)):
        hl.hm[1].bias.copy_(hm_floor - z0)
    return model


def synth_detector(voxel_size, seed=0, sweeps=1, gain='preserve', second_stage=False):
    """Seeded random-init CenterPoint of the reference architecture (there are no checkpoints offline;
    sweeps > 1: the multi-sweep configuration - DynamicMeanVFE on 6 point features, centerpoint_3sweeps.yaml;
    second_stage: the centerpoint_pdv_3sweeps shape with its PDVHead): BatchNorm running statistics randomised so
    BN folding is exercised, then one of two weight sets:
      gain='preserve' (default since round 6): `variance_preserving_init` - activations stay O(1) through all 46 layers and the final
                      boxes sit on the frame's points (a few hundred per frame, different for every frame);
      gain='default'  (rounds 1-5, kept for continuity): the default initialisers, final-conv biases of hm / dim / iou spread so that a
                      few hundred boxes pass SCORE_THRESH - the data-dependent signal dies in the dense stage
                      and the boxes are the zero-padding border's, the same for every frame.
    Returns (model on CPU in eval mode, cfg, dataset_info)."""
    from .centerpoint import SyntheticDatasetInfo, build_network
    from .config import centerpoint_1sweep_cfg, centerpoint_3sweeps_cfg, centerpoint_pdv_cfg
    if gain not in ('preserve', 'default'):
        raise DetZeroHipError('synth_detector: gain is "preserve" or "default"')
    if second_stage:
        cfg = centerpoint_pdv_cfg(tuple(voxel_size))
    else:
        cfg = centerpoint_1sweep_cfg(tuple(voxel_size)) if sweeps == 1 else centerpoint_3sweeps_cfg(tuple(voxel_size))
    info = SyntheticDatasetInfo(cfg, num_point_features=5 if (sweeps == 1 and not second_stage) else 6)
    torch.manual_seed(seed)
    model = build_network(cfg.MODEL, len(cfg.CLASS_NAMES), info).eval()
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        if not second_stage or gain == 'preserve':
            for m in model.modules():
                if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)) and not _in_roi_head(model, m):
                    m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
                    m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
                    m.weight.data.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                    m.bias.data.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)
        if gain == 'preserve':
            variance_preserving_init(model, seed)
        else:
            hl = model.dense_head.heads_list[0]
            hl.hm[1].bias.fill_(-0.5)
            hl.dim[1].bias.copy_(torch.tensor([1.2, 0.6, 0.4]))
            hl.iou[1].bias.fill_(0.6)
    return model, cfg, info


def _in_roi_head(model, module):
    rh = getattr(model, 'roi_head', None)
    return rh is not None and any(module is m for m in rh.modules())
