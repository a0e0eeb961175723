"""Generate golden vectors by running the REFERENCE's own Python modules (imported from
/root/reference, CPU torch) on seeded inputs.  Run in the build container only:

    python tests/golden/gen_golden.py

The reference cannot travel to the GPU box, so the resulting small .npz fixtures are committed next
to this script; tests/test_oracle_golden.py pins the oracle to them and the -m gpu tests pin the
HIP path to the oracle and to the same fixtures.

What is imported from the reference (files loaded individually, package __init__ files that pull in
spconv / the CUDA extensions are bypassed):
  detection/detzero_det/models/centerpoint_modules/vfe.py            MeanVFE, DynamicMeanVFE
  detection/detzero_det/models/centerpoint_modules/backbone2d.py     BaseBEVBackbone
  detection/detzero_det/models/centerpoint_modules/center_head.py    CenterHead (SeparateHead, generate_predicted_boxes)
  detection/detzero_det/utils/centernet_utils.py, model_nms_utils.py decode, class-agnostic NMS
  refining/detzero_refine/models/modules/transformer/multi_head_attention.py MultiheadAttention
Stubs (not part of the arithmetic under test, or unavailable here):
  numba.jit (decorator only), torch_scatter.scatter_mean (index_add / count in float64, rounded),
  Tensor.cuda() -> identity (no GPU in the container),
  iou3d_nms_utils.nms_gpu -> oracle/_ref build of the reference's own iou3d_cpu.cpp (rotated IoU)
  driving the sweep of iou3d_nms.cpp:145-156.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def install_stubs():
    from oracle import refbuild
    torch.Tensor.cuda = lambda self, *a, **k: self
    _mod('numba', jit=lambda *a, **k: (lambda f: f))

    def scatter_mean(src, index, dim=0):
        n = int(index.max().item()) + 1
        out = torch.zeros((n, src.shape[1]), dtype=torch.float64)
        out.index_add_(0, index, src.double())
        cnt = torch.bincount(index, minlength=n).double().clamp(min=1)
        return (out / cnt[:, None]).float()
    _mod('torch_scatter', scatter_mean=scatter_mean)

    def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
        order = scores.sort(0, descending=True)[1]
        if pre_maxsize is not None:
            order = order[:pre_maxsize]
        b = boxes[order].contiguous().numpy()
        keep = refbuild.nms_with_reference_iou(b, thresh)
        return order[torch.from_numpy(keep)].contiguous(), None
    _mod('detzero_utils')
    _mod('detzero_utils.common_utils')
    _mod('detzero_utils.box_utils', boxes_to_corners_3d=None)
    _mod('detzero_utils.ops')
    _mod('detzero_utils.ops.iou3d_nms')
    sys.modules['detzero_utils.ops.iou3d_nms'].iou3d_nms_utils = _mod('detzero_utils.ops.iou3d_nms.iou3d_nms_utils',
                                                                      nms_gpu=nms_gpu)
    sys.modules['detzero_utils'].box_utils = sys.modules['detzero_utils.box_utils']
    sys.modules['detzero_utils'].common_utils = sys.modules['detzero_utils.common_utils']
    _mod('detzero_det')
    utils = _mod('detzero_det.utils')
    det = REF + '/detection/detzero_det'
    utils.centernet_utils = _load('detzero_det.utils.centernet_utils', det + '/utils/centernet_utils.py')
    utils.model_nms_utils = _load('detzero_det.utils.model_nms_utils', det + '/utils/model_nms_utils.py')

    class _Loss(nn.Module):
        pass
    utils.loss_utils = _mod('detzero_det.utils.loss_utils', FocalLossCenterNet=_Loss, RegLossCenterNet=_Loss)
    mods = {}
    for f in ('vfe', 'backbone2d', 'center_head'):
        mods[f] = _load('ref_' + f, det + '/models/centerpoint_modules/%s.py' % f)
    return mods


def randomize_bn(module, gen):
    for m in module.modules():
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)


def sd_np(module, prefix):
    return {prefix + k: v.detach().numpy() for k, v in module.state_dict().items()}


def main():
    from detzero_amd.config import AttrDict
    from detzero_amd.synth import synth_waymo_frame, POINT_CLOUD_RANGE, VOXEL_SIZE_02
    from oracle import voxelize as ov
    mods = install_stubs()
    gen = torch.Generator().manual_seed(1234)
    out = {}

    # ---- MeanVFE (vfe.py:66-83) on a hard-voxelized 20k frame (BASELINE configs[0])
    pts = synth_waymo_frame(3, 20000)
    pts = pts[ov.mask_points_by_range(pts, POINT_CLOUD_RANGE)]
    voxels, coords, nump = ov.hard_voxelize(pts, POINT_CLOUD_RANGE, VOXEL_SIZE_02, 5, 200000)
    sel = np.arange(0, voxels.shape[0], 7)[:1500]
    vfe = mods['vfe'].MeanVFE(None, 5)
    bd = vfe({'voxels': torch.from_numpy(voxels[sel]), 'voxel_num_points': torch.from_numpy(nump[sel]).float()})
    out['meanvfe_voxels'] = voxels[sel]
    out['meanvfe_num'] = nump[sel]
    out['meanvfe_out'] = bd['voxel_features'].numpy()

    # ---- DynamicMeanVFE (vfe.py:109-147), batch of 2 frames, 6000 points
    p0 = synth_waymo_frame(5, 3000); p1 = synth_waymo_frame(6, 3000)
    edge = np.array([[75.2, 0, 0, .5, .5], [-75.2, -75.2, -2, .5, .5], [0, 0, 4.0, .1, .1], [75.19999, 75.2, 3.99, .2, .2]], np.float32)
    p0 = np.concatenate([p0, edge], 0)
    pb = np.concatenate([np.concatenate([np.zeros((p0.shape[0], 1), np.float32), p0], 1),
                         np.concatenate([np.ones((p1.shape[0], 1), np.float32), p1], 1)], 0)
    grid = ov.grid_size_of(POINT_CLOUD_RANGE, VOXEL_SIZE_02)
    dvfe = mods['vfe'].DynamicMeanVFE(None, 5, VOXEL_SIZE_02, [int(g) for g in grid], POINT_CLOUD_RANGE)
    bd = dvfe({'batch_size': 2, 'points': torch.from_numpy(pb)})
    out['dynvfe_points'] = pb
    out['dynvfe_feats'] = bd['voxel_features'].numpy()
    out['dynvfe_coords'] = bd['voxel_coords'].numpy().astype(np.int32)

    # ---- BaseBEVBackbone (backbone2d.py), reduced channel config, 24x24 map, batch 2
    bcfg = AttrDict({'LAYER_NUMS': [2, 2], 'LAYER_STRIDES': [1, 2], 'NUM_FILTERS': [32, 64],
                     'UPSAMPLE_STRIDES': [1, 2], 'NUM_UPSAMPLE_FILTERS': [32, 32]})
    bb = mods['backbone2d'].BaseBEVBackbone(bcfg, 32).eval()
    randomize_bn(bb, gen)
    x = torch.randn((2, 32, 24, 24), generator=gen)
    with torch.no_grad():
        y = bb({'spatial_features': x})['spatial_features_2d']
    out.update({'bev_' + k: v for k, v in sd_np(bb, 'backbone2d.').items()})
    out['bev_in'] = x.numpy(); out['bev_out'] = y.numpy()

    # ---- CenterHead convs + generate_predicted_boxes (center_head.py), 64 ch in, shared 32
    hcfg = AttrDict({
        'CLASS_NAMES_EACH_HEAD': [['Vehicle', 'Pedestrian', 'Cyclist']], 'SHARED_CONV_CHANNEL': 32,
        'USE_BIAS_BEFORE_NORM': True, 'NUM_HM_CONV': 2, 'IOU_WEIGHT': 1,
        'SEPARATE_HEAD_CFG': {'HEAD_ORDER': ['center', 'center_z', 'dim', 'rot', 'iou'],
                              'HEAD_DICT': {'center': {'out_channels': 2, 'num_conv': 2}, 'center_z': {'out_channels': 1, 'num_conv': 2},
                                            'dim': {'out_channels': 3, 'num_conv': 2}, 'rot': {'out_channels': 2, 'num_conv': 2},
                                            'iou': {'out_channels': 1, 'num_conv': 2}}},
        'TARGET_ASSIGNER_CONFIG': {'FEATURE_MAP_STRIDE': 8},
        'POST_PROCESSING': {'SCORE_THRESH': 0.03, 'POST_CENTER_LIMIT_RANGE': [-80, -80, -10.0, 80, 80, 10.0],
                            'MAX_OBJ_PER_SAMPLE': 100,
                            'NMS_CONFIG': {'NMS_TYPE': 'nms_gpu', 'NMS_THRESH': 0.7, 'NMS_PRE_MAXSIZE': 4096, 'NMS_POST_MAXSIZE': 500}},
    })
    names = ['Vehicle', 'Pedestrian', 'Cyclist']
    head = mods['center_head'].CenterHead(hcfg, 64, 3, names, grid, POINT_CLOUD_RANGE, VOXEL_SIZE_02).eval()
    randomize_bn(head, gen)
    # spread the final-conv biases so that decoded boxes are non-degenerate and overlap
    hl = head.heads_list[0]
    with torch.no_grad():
        hl.hm[1].bias.fill_(-0.5)
        hl.dim[1].bias.copy_(torch.tensor([1.2, 0.6, 0.4]))
        hl.iou[1].bias.fill_(0.6)
    x = torch.randn((2, 64, 24, 24), generator=gen) * 0.5
    with torch.no_grad():
        s = head.shared_conv(x)
        pd = hl(s)
        boxes = head.generate_predicted_boxes(2, [pd])
    out.update({'head_' + k: v for k, v in sd_np(head, 'dense_head.').items()})
    out['head_in'] = x.numpy()
    for k, v in pd.items():
        out['head_pred_' + k] = v.numpy()
    for i, bdict in enumerate(boxes):
        out['head_boxes_%d' % i] = bdict['pred_boxes'].numpy()
        out['head_scores_%d' % i] = bdict['pred_scores'].numpy()
        out['head_labels_%d' % i] = bdict['pred_labels'].numpy()
        print('frame', i, 'final boxes', bdict['pred_boxes'].shape[0])
    # decode-only (before NMS) from the reference's centernet_utils
    cu = sys.modules['detzero_det.utils.centernet_utils']
    with torch.no_grad():
        dec = cu.decode_bbox_from_heatmap(
            heatmap=pd['hm'].sigmoid(), rot_cos=pd['rot'][:, 0:1], rot_sin=pd['rot'][:, 1:2], center=pd['center'],
            center_z=pd['center_z'], dim=pd['dim'].exp(), vel=None, batch_iou=pd['iou'],
            point_cloud_range=POINT_CLOUD_RANGE, voxel_size=VOXEL_SIZE_02, feature_map_stride=8, K=100,
            circle_nms=False, score_thresh=0.03, post_center_limit_range=torch.tensor([-80, -80, -10.0, 80, 80, 10.0]))
    for i, d in enumerate(dec):
        out['dec_boxes_%d' % i] = d['pred_boxes'].numpy()
        out['dec_scores_%d' % i] = d['pred_scores'].numpy()
        out['dec_labels_%d' % i] = d['pred_labels'].numpy()
        print('frame', i, 'decoded', d['pred_boxes'].shape[0])

    np.savez_compressed(os.path.join(HERE, 'det_golden.npz'), **out)

    # ---- rotated IoU: the reference's own C++ (iou3d_cpu.cpp via oracle/_ref)
    from detzero_amd.synth import synth_boxes
    from oracle import refbuild
    a = synth_boxes(11, 96)
    rng = np.random.default_rng(12)
    b = a[:80].copy()                       # every b overlaps its a: shifts, rescales and rotations of all sizes
    b[20:, :2] += rng.normal(0, 0.6, size=(60, 2)).astype(np.float32)
    b[30:, 3:5] *= rng.uniform(0.7, 1.4, size=(50, 2)).astype(np.float32)
    b[40:, 6] += rng.uniform(-1.6, 1.6, size=40).astype(np.float32)
    a[90:96] = a[0]                         # a few boxes sharing one place (cluster)
    a[90:96, 6] += np.linspace(0.0, 1.5, 6).astype(np.float32)
    iou = refbuild.boxes_iou_bev_reference(a, b)
    np.savez_compressed(os.path.join(HERE, 'iou_golden.npz'), a=a, b=b, iou=iou)
    print('iou golden', iou.shape, float(iou.max()))

    # ---- refiner attention (multi_head_attention.py) : self-contained nn.Module
    mha_mod = _load('ref_mha', REF + '/refining/detzero_refine/models/modules/transformer/multi_head_attention.py')
    mha = mha_mod.MultiheadAttention(64, 2, dropout=0.0).eval()
    q = torch.randn((5, 3, 64), generator=gen); k = torch.randn((40, 3, 64), generator=gen); v = torch.randn((40, 3, 64), generator=gen)
    kpm = torch.zeros((3, 40), dtype=torch.bool); kpm[1, 25:] = True; kpm[2, 3:] = True
    with torch.no_grad():
        o, _ = mha(q, k, v, key_padding_mask=kpm)
    sdm = {('mha_' + kk): vv.detach().numpy() for kk, vv in mha.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, 'mha_golden.npz'), q=q.numpy(), k=k.numpy(), v=v.numpy(), kpm=kpm.numpy(),
                        out=o.numpy(), **sdm)
    print('mha golden', o.shape)


if __name__ == '__main__':
    main()
