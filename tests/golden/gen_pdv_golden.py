"""Golden vectors of the PDV second stage from the REFERENCE's own Python classes, CPU.

    python tests/golden/gen_pdv_golden.py      (build container only; needs /root/reference)  ->  tests/golden/pdv_golden.npz

Imported as they are from /root/reference: pdv_head.PDVHead (with RoIHeadTemplate), pointnet2_stack.pointnet2_modules
(StackSAModuleMSGAttention) and pointnet2_utils (QueryAndGroup, BallQueryCount, GroupingOperation), kde_utils, attention_utils,
density_utils, voxel_aggregation_utils, box_coder_utils, common_utils.  Replaced: the three compiled CUDA extensions by the
numpy kernels of oracle/pdv.py (pointnet2_stack_cuda.{ball_query_count_wrapper, group_points_wrapper},
roiaware_pool3d_cuda.points_in_multi_boxes_gpu), torch.cuda.{Int,Float}Tensor by their CPU types, the loss / target-assignment
classes (training only) by empty modules.  The sparse tensors x_conv3 / x_conv4 are small objects with the four attributes
PDVHead reads (.indices, .features, .spatial_shape, .batch_size); their active sets come from the oracle's index builder on the
frame's voxels, their features are seeded noise.  Weights: detzero_amd.synth.synth_state_dict (a function of key, shape, seed),
so the fixture stores none.
"""
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
sys.path.insert(0, HERE)

RANGE = np.array([-24.0, -24.0, -2.0, 24.0, 24.0, 4.0], np.float32)
VOXEL = [0.1, 0.1, 0.15]
WEIGHT_SEED = 4
ROI_SUBSET = [0, 3, 9, 13, 17, 27]


def roi_head_cfg():
    from detzero_amd.config import AttrDict
    return AttrDict({
        'NAME': 'PDVHead', 'CLASS_AGNOSTIC': True, 'SHARED_FC': [256, 256], 'CLS_FC': [256, 256], 'REG_FC': [256, 256], 'DP_RATIO': 0.3,
        'NMS_CONFIG': {'TEST': {'NMS_TYPE': 'nms_gpu', 'MULTI_CLASSES_NMS': False, 'NMS_PRE_MAXSIZE': 1024, 'NMS_POST_MAXSIZE': 512, 'NMS_THRESH': 0.7}},
        'VOXEL_AGGREGATION': {'NUM_FEATURES': [64, 128], 'FEATURE_LOCATIONS': ['x_conv3', 'x_conv4']},
        'ROI_GRID_POOL': {
            'FEATURE_LOCATIONS': ['x_conv3', 'x_conv4'], 'GRID_SIZE': 6,
            'POOL_LAYERS': {'x_conv3': {'MLPS': [[32, 32], [32, 32]], 'POOL_RADIUS': [0.8, 1.2], 'NSAMPLE': [16, 16], 'POOL_METHOD': 'max_pool', 'USE_DENSITY': True},
                            'x_conv4': {'MLPS': [[64, 64], [64, 64]], 'POOL_RADIUS': [1.2, 2.4], 'NSAMPLE': [16, 16], 'POOL_METHOD': 'max_pool', 'USE_DENSITY': True}},
            'ATTENTION': {'ENABLED': True, 'NUM_FEATURES': 192, 'NUM_HEADS': 1, 'NUM_HIDDEN_FEATURES': 128, 'NUM_LAYERS': 1,
                          'POSITIONAL_ENCODER': 'density_grid_points', 'MAX_NUM_BOXES': 20, 'DROPOUT': 0.1, 'COMBINE': True, 'MASK_EMPTY_POINTS': True}},
        'TARGET_CONFIG': {'BOX_CODER': 'ResidualCoder', 'ROI_PER_IMAGE': 128, 'FG_RATIO': 0.5},
        'LOSS_CONFIG': {'CLS_LOSS': 'BinaryCrossEntropy', 'REG_LOSS': 'smooth-l1', 'CORNER_LOSS_REGULARIZATION': True,
                        'LOSS_WEIGHTS': {'rcnn_cls_weight': 1.0, 'rcnn_reg_weight': 1.0, 'rcnn_corner_weight': 1.0, 'code_weights': [1.0] * 7}},
    })


def scene(seed=0):
    """Two frames inside RANGE: points (N, 1+5) [b,x,y,z,i,e], x_conv3 / x_conv4 active sets + seeded features, padded rois."""
    from detzero_amd.synth import synth_waymo_frame
    from oracle import sparse as osp, voxelize as ov
    rng = np.random.default_rng(seed)
    pts_b, c3, c4, rois = [], [], [], []
    n_roi = [14, 11]
    for b in range(2):
        p = synth_waymo_frame(50 + b, 60000)
        m = np.all((p[:, :3] > RANGE[:3] + 0.05) & (p[:, :3] < RANGE[3:] - 0.05), axis=1)
        p = p[m][:9000]
        pts_b.append(np.concatenate([np.full((p.shape[0], 1), b, np.float32), p], 1))
        _, czyx, _ = ov.hard_voxelize(p, RANGE, VOXEL, 5, 200000)
        coords = np.concatenate([np.zeros((czyx.shape[0], 1), np.int32), czyx], 1)
        shape = [41, 480, 480]
        coords = coords[osp.canonical_order(coords, shape)]
        K3, S2 = (3, 3, 3), (2, 2, 2)
        l2, s2 = osp.conv_out_coords(coords, shape, K3, S2, (1, 1, 1))
        l3, s3 = osp.conv_out_coords(l2, s2, K3, S2, (1, 1, 1))
        l4, s4 = osp.conv_out_coords(l3, s3, K3, S2, (0, 1, 1))
        l3, l4 = l3.copy(), l4.copy()
        l3[:, 0] = b; l4[:, 0] = b
        c3.append(l3); c4.append(l4)
        centres = p[rng.choice(p.shape[0], n_roi[b], replace=False), :3]
        r = np.zeros((14, 7), np.float32)
        r[:n_roi[b], :3] = centres + rng.normal(0, 0.2, (n_roi[b], 3))
        r[:n_roi[b], 3:6] = rng.uniform([1.5, 0.8, 1.2], [5.0, 2.4, 2.2], (n_roi[b], 3))
        r[:n_roi[b], 6] = rng.uniform(-np.pi, np.pi, n_roi[b])
        rois.append(r)
    c3, c4 = np.concatenate(c3), np.concatenate(c4)
    f3 = rng.standard_normal((c3.shape[0], 64)).astype(np.float32)
    f4 = rng.standard_normal((c4.shape[0], 128)).astype(np.float32)
    scores = rng.uniform(0.1, 0.9, (2, 14)).astype(np.float32)
    labels = rng.integers(1, 4, (2, 14)).astype(np.int64)
    for b in range(2):
        scores[b, n_roi[b]:] = 0; labels[b, n_roi[b]:] = 0
    return {'points': np.concatenate(pts_b), 'c3': c3.astype(np.int32), 'f3': f3, 's3': np.array(s3), 'c4': c4.astype(np.int32), 'f4': f4,
            's4': np.array(s4), 'rois': np.stack(rois), 'roi_scores': scores, 'roi_labels': labels}


RANGE_BIG = np.array([-75.2, -75.2, -2.0, 75.2, 75.2, 4.0], np.float32)
N_ROI_BIG = 320


def scene_big(seed=0):
    """The bench shape (tools/bench_pdv.py: a full-range frame, hundreds of RoIs): one 120k-point frame over the whole Waymo range,
    320 RoIs on the point cloud.  Everything is a function of the seed, so the fixture stores outputs only."""
    from detzero_amd.synth import synth_waymo_frame
    from oracle import sparse as osp, voxelize as ov
    rng = np.random.default_rng(1000 + seed)
    p = synth_waymo_frame(90 + seed, 120000)
    p = p[np.all((p[:, :3] > RANGE_BIG[:3] + 0.05) & (p[:, :3] < RANGE_BIG[3:] - 0.05), axis=1)]
    pts = np.concatenate([np.zeros((p.shape[0], 1), np.float32), p], 1)
    _, czyx, _ = ov.hard_voxelize(p, RANGE_BIG, VOXEL, 5, 200000)
    coords = np.concatenate([np.zeros((czyx.shape[0], 1), np.int32), czyx], 1)
    shape = [41, 1504, 1504]
    coords = coords[osp.canonical_order(coords, shape)]
    K3, S2 = (3, 3, 3), (2, 2, 2)
    l2, s2 = osp.conv_out_coords(coords, shape, K3, S2, (1, 1, 1))
    l3, s3 = osp.conv_out_coords(l2, s2, K3, S2, (1, 1, 1))
    l4, s4 = osp.conv_out_coords(l3, s3, K3, S2, (0, 1, 1))
    centres = p[rng.choice(p.shape[0], N_ROI_BIG, replace=False), :3]
    r = np.zeros((1, N_ROI_BIG, 7), np.float32)
    r[0, :, :3] = centres + rng.normal(0, 0.2, (N_ROI_BIG, 3))
    r[0, :, 3:6] = rng.uniform([1.5, 0.8, 1.2], [5.0, 2.4, 2.2], (N_ROI_BIG, 3))
    r[0, :, 6] = rng.uniform(-np.pi, np.pi, N_ROI_BIG)
    f3 = rng.standard_normal((l3.shape[0], 64)).astype(np.float32)
    f4 = rng.standard_normal((l4.shape[0], 128)).astype(np.float32)
    return {'points': pts, 'c3': l3.astype(np.int32), 'f3': f3, 's3': np.array(s3), 'c4': l4.astype(np.int32), 'f4': f4, 's4': np.array(s4),
            'rois': r, 'roi_scores': rng.uniform(0.1, 0.9, (1, N_ROI_BIG)).astype(np.float32), 'roi_labels': rng.integers(1, 4, (1, N_ROI_BIG)).astype(np.int64)}


def install_reference():
    import gen_golden as gg
    gg.install_stubs()                                   # numba, torch_scatter, iou3d_nms_utils.nms_gpu, detzero_det.utils.{centernet,model_nms}_utils
    from oracle import pdv as opdv
    torch.cuda.IntTensor = torch.IntTensor
    torch.cuda.FloatTensor = torch.FloatTensor
    du = sys.modules['detzero_utils']
    du.__path__ = []
    du.common_utils = gg._load('detzero_utils.common_utils', REF + '/utils/detzero_utils/common_utils.py')
    du.kde_utils = gg._load('detzero_utils.kde_utils', REF + '/utils/detzero_utils/kde_utils.py')

    def ball_query_count_wrapper(b, m, radius, nsample, new_xyz, new_cnt, xyz, cnt, idx):
        idx.copy_(torch.from_numpy(opdv.ball_query_count(radius, nsample, xyz.numpy(), cnt.numpy(), new_xyz.numpy(), new_cnt.numpy())))

    def group_points_wrapper(b, m, c, nsample, features, fcnt, idx, icnt, out):
        out.copy_(torch.from_numpy(opdv.group_points(features.numpy(), fcnt.numpy(), idx.numpy(), icnt.numpy())))

    def points_in_multi_boxes_gpu(boxes, pts, out, max_num_boxes):
        out.copy_(torch.from_numpy(opdv.points_in_multi_boxes(pts.numpy(), boxes.numpy(), max_num_boxes)))
        return 1
    stack_pkg = 'detzero_utils.ops.pointnet2.pointnet2_stack'
    for name in ('detzero_utils.ops.pointnet2', stack_pkg, 'detzero_utils.ops.roiaware_pool3d'):
        gg._mod(name).__path__ = []
    cuda_ext = gg._mod(stack_pkg + '.pointnet2_stack_cuda', ball_query_count_wrapper=ball_query_count_wrapper, group_points_wrapper=group_points_wrapper)
    sys.modules[stack_pkg].pointnet2_stack_cuda = cuda_ext
    roi_ext = gg._mod('detzero_utils.ops.roiaware_pool3d.roiaware_pool3d_cuda', points_in_multi_boxes_gpu=points_in_multi_boxes_gpu)
    sys.modules['detzero_utils.ops.roiaware_pool3d'].roiaware_pool3d_cuda = roi_ext
    ops_dir = REF + '/utils/detzero_utils/ops'
    sys.modules['detzero_utils.ops.roiaware_pool3d'].roiaware_pool3d_utils = gg._load('detzero_utils.ops.roiaware_pool3d.roiaware_pool3d_utils', ops_dir + '/roiaware_pool3d/roiaware_pool3d_utils.py')
    sys.modules[stack_pkg].pointnet2_utils = gg._load(stack_pkg + '.pointnet2_utils', ops_dir + '/pointnet2/pointnet2_stack/pointnet2_utils.py')
    sys.modules[stack_pkg].pointnet2_modules = gg._load(stack_pkg + '.pointnet2_modules', ops_dir + '/pointnet2/pointnet2_stack/pointnet2_modules.py')
    det = REF + '/detection/detzero_det'
    utils = sys.modules['detzero_det.utils']
    for name in ('box_coder_utils', 'voxel_aggregation_utils', 'density_utils', 'attention_utils'):
        setattr(utils, name, gg._load('detzero_det.utils.' + name, det + '/utils/%s.py' % name))

    class _Empty(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
    utils.loss_utils.WeightedSmoothL1Loss = _Empty
    models = gg._mod('detzero_det.models'); models.__path__ = []
    cpm = gg._mod('detzero_det.models.centerpoint_modules'); cpm.__path__ = []
    gg._mod('detzero_det.models.centerpoint_modules.proposal_target_layer', ProposalTargetLayer=_Empty)
    return gg._load('detzero_det.models.centerpoint_modules.pdv_head', det + '/models/centerpoint_modules/pdv_head.py')


class SparseStub:
    def __init__(self, indices, features, spatial_shape, batch_size):
        self.indices, self.features, self.spatial_shape, self.batch_size = indices, features, list(spatial_shape), batch_size


def main():
    from detzero_amd.synth import synth_state_dict
    pdv = install_reference()
    sc = scene()
    head = pdv.PDVHead(512, roi_head_cfg(), RANGE, VOXEL, num_class=1).eval()
    head.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in head.state_dict().items()}, seed=WEIGHT_SEED), strict=True)
    t = torch.from_numpy
    bd = {'batch_size': 2, 'points': t(sc['points']), 'rois': t(sc['rois']), 'roi_scores': t(sc['roi_scores']), 'roi_labels': t(sc['roi_labels']),
          'has_class_labels': True, 'multi_scale_3d_strides': {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8},
          'multi_scale_3d_features': {'x_conv3': SparseStub(t(sc['c3']), t(sc['f3']), sc['s3'], 2), 'x_conv4': SparseStub(t(sc['c4']), t(sc['f4']), sc['s4'], 2)}}
    out = {k: v for k, v in sc.items()}
    out['manifest_keys'] = np.array(list(head.state_dict().keys()))
    out['manifest_shapes'] = np.array([str(tuple(v.shape)) for v in head.state_dict().values()])
    with torch.no_grad():
        pf, pc = head.get_point_voxel_features(bd)
        for loc in ('x_conv3', 'x_conv4'):
            out['pc_' + loc], out['pf_' + loc] = pc[loc].numpy(), pf[loc].numpy()
        bd['point_features'], bd['point_coords'] = pf, pc
        pooled, g_pts, l_pts, ball = head.roi_grid_pool(bd)
        out['pooled'], out['grid_global'], out['grid_local'], out['ball_idxs'] = pooled.numpy(), g_pts.numpy(), l_pts.numpy(), ball.numpy()
        pos = head.get_positional_input(bd['points'], bd['rois'], l_pts)
        out['positional_input'] = pos.numpy()
        mask = (ball == 0).all(-1)
        out['key_padding_mask'] = mask.numpy()
        att = head.attention_head(pooled, pos, mask)
        out['attention'] = att.numpy()
        res = head({k: v for k, v in bd.items() if k not in ('point_features', 'point_coords')})
        out['batch_cls_preds'], out['batch_box_preds'] = res['batch_cls_preds'].numpy(), res['batch_box_preds'].numpy()
    print({k: (v.shape if hasattr(v, 'shape') else v) for k, v in out.items() if k not in ('manifest_keys', 'manifest_shapes')})
    print('masked grid points: %.1f %%, empty rois %d' % (100 * out['key_padding_mask'].mean(), int(out['key_padding_mask'].all(-1).sum())))
    keep = ['points', 'c3', 'f3', 's3', 'c4', 'f4', 's4', 'rois', 'roi_scores', 'roi_labels', 'manifest_keys', 'manifest_shapes', 'pc_x_conv3', 'pc_x_conv4',
            'grid_local', 'ball_idxs', 'positional_input', 'key_padding_mask', 'batch_cls_preds', 'batch_box_preds']
    small = {k: out[k] for k in keep}
    small['pf_x_conv3_head'], small['pf_x_conv4_head'] = out['pf_x_conv3'][:64], out['pf_x_conv4'][:64]
    small['roi_subset'] = np.array(ROI_SUBSET)                     # pooled / attended features of a few RoIs (one of them padding)
    small['pooled'] = out['pooled'][ROI_SUBSET].astype(np.float32)
    small['attention'] = out['attention'][ROI_SUBSET].astype(np.float32)
    small['ball_idxs'] = out['ball_idxs'].astype(np.int16)
    np.savez_compressed(os.path.join(HERE, 'pdv_golden.npz'), **small)
    print('saved', os.path.getsize(os.path.join(HERE, 'pdv_golden.npz')) // 1024, 'KiB')


def main_big():
    """pdv_big_golden.npz: the reference's PDVHead over scene_big() - final boxes / confidences of all 320 RoIs, the key-padding
    mask, and the pooled / attended features of a few RoIs."""
    import time
    from detzero_amd.synth import synth_state_dict
    pdv = install_reference()
    sc = scene_big()
    head = pdv.PDVHead(512, roi_head_cfg(), RANGE_BIG, VOXEL, num_class=1).eval()
    head.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in head.state_dict().items()}, seed=WEIGHT_SEED), strict=True)
    t = torch.from_numpy
    bd = {'batch_size': 1, 'points': t(sc['points']), 'rois': t(sc['rois']), 'roi_scores': t(sc['roi_scores']), 'roi_labels': t(sc['roi_labels']),
          'has_class_labels': True, 'multi_scale_3d_strides': {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8},
          'multi_scale_3d_features': {'x_conv3': SparseStub(t(sc['c3']), t(sc['f3']), sc['s3'], 1), 'x_conv4': SparseStub(t(sc['c4']), t(sc['f4']), sc['s4'], 1)}}
    t0 = time.time()
    with torch.no_grad():
        pf, pc = head.get_point_voxel_features(bd)
        bd['point_features'], bd['point_coords'] = pf, pc
        pooled, g_pts, l_pts, ball = head.roi_grid_pool(bd)
        pos = head.get_positional_input(bd['points'], bd['rois'], l_pts)
        mask = (ball == 0).all(-1)
        att = head.attention_head(pooled, pos, mask)
        res = head({k: v for k, v in bd.items() if k not in ('point_features', 'point_coords')})
    sub = np.array([0, 211])
    small = {'n_points': np.array(sc['points'].shape[0]), 'n_c3': np.array(sc['c3'].shape[0]), 'n_c4': np.array(sc['c4'].shape[0]),
             'n_centroids3': np.array(pc['x_conv3'].shape[0]), 'n_centroids4': np.array(pc['x_conv4'].shape[0]),
             'key_padding_mask': np.packbits(mask.numpy()), 'roi_subset': sub, 'pooled': pooled.numpy()[sub].astype(np.float32),
             'attention': att.numpy()[sub].astype(np.float32), 'ball_row_sums': ball.numpy().astype(np.int64).sum(axis=(1, 2)),
             'batch_cls_preds': res['batch_cls_preds'].numpy(), 'batch_box_preds': res['batch_box_preds'].numpy()}
    print('reference PDVHead on %d points, %d RoIs: %.0f s; masked grid points %.1f %%' % (sc['points'].shape[0], N_ROI_BIG, time.time() - t0, 100 * mask.numpy().mean()))
    np.savez_compressed(os.path.join(HERE, 'pdv_big_golden.npz'), **small)
    print('saved', os.path.getsize(os.path.join(HERE, 'pdv_big_golden.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    if '--big' in sys.argv:
        main_big()
    else:
        main()
