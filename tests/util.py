"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np
import torch
import torch.nn as nn

from detzero_amd.centerpoint import SyntheticDatasetInfo, build_network
from detzero_amd.config import centerpoint_1sweep_cfg
from detzero_amd.synth import POINT_CLOUD_RANGE, synth_waymo_frame

POST = {'SCORE_THRESH': 0.03, 'POST_CENTER_LIMIT_RANGE': [-80, -80, -10.0, 80, 80, 10.0], 'MAX_OBJ_PER_SAMPLE': 500,
        'NMS_THRESH': 0.7, 'NMS_PRE_MAXSIZE': 4096, 'NMS_POST_MAXSIZE': 500}


def make_model(voxel_size, seed=0, sweeps=1, gain='preserve', second_stage=False):
    """Seeded random-init CenterPoint (reference architecture) - see detzero_amd.centerpoint.synth_detector.  gain='preserve' (default):
    the variance-preserving weight set whose boxes depend on the frame; 'default': the rounds 1-5 set (frame-independent border boxes)."""
    from detzero_amd.centerpoint import synth_detector
    return synth_detector(voxel_size, seed, sweeps, gain=gain, second_stage=second_stage)


def cpu_state_dict(model):
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


def masked_frame(seed, n):
    pts = synth_waymo_frame(seed, n)
    m = (pts[:, 0] >= POINT_CLOUD_RANGE[0]) & (pts[:, 0] <= POINT_CLOUD_RANGE[3]) & \
        (pts[:, 1] >= POINT_CLOUD_RANGE[1]) & (pts[:, 1] <= POINT_CLOUD_RANGE[4])
    return pts[m]


def oracle_detect(sd, points, info, post=POST, dynamic=False, times=None):
    """Whole per-frame path on the CPU oracle: returns dict of intermediates + final boxes.
    dynamic: DynamicMeanVFE (multi-sweep configs, vfe.py:109-147) instead of the hard voxelizer + MeanVFE.
    times (optional dict): seconds per stage are ADDED to its keys voxelize / sparse_backbone / dense / post (bench.py's cpu_baseline)."""
    import time
    from oracle import dense, sparse as osp, voxelize as ov
    t0 = time.perf_counter()
    if dynamic:
        pb = np.concatenate([np.zeros((points.shape[0], 1), np.float32), points], 1)
        feats, coords = ov.dynamic_mean_vfe(pb, info.point_cloud_range, info.voxel_size)
        vox = nump = None
    else:
        vox, czyx, nump = ov.hard_voxelize(points, info.point_cloud_range, info.voxel_size, 5, info.max_voxels['test'])
        feats = ov.mean_vfe(vox, nump)
        coords = np.concatenate([np.zeros((czyx.shape[0], 1), np.int32), czyx], 1)
    t1 = time.perf_counter()
    grid = info.grid_size
    sparse_shape = [int(grid[2]) + 1, int(grid[1]), int(grid[0])]
    res = osp.backbone_forward(sd, feats, coords, sparse_shape)
    t2 = time.perf_counter()
    x, oc, shape = res['encoded']
    bev = osp.to_bev(x, oc, shape, 1)
    f2d = dense.bev_backbone_forward(sd, bev)
    pred = dense.center_head_forward(sd, f2d)
    t3 = time.perf_counter()
    final = dense.generate_predicted_boxes(pred, info.point_cloud_range, info.voxel_size, 8, post)
    t4 = time.perf_counter()
    if times is not None:
        for k, v in (('voxelize', t1 - t0), ('sparse_backbone', t2 - t1), ('dense', t3 - t2), ('post', t4 - t3)):
            times[k] = times.get(k, 0.0) + v
    return {'voxels': vox, 'coords': coords, 'num_points': nump, 'feats': feats, 'backbone': res, 'bev': bev,
            'f2d': f2d, 'pred': pred, 'final': final}


def oracle_features_f64(sd, points, info, dynamic=False):
    """The feature path of `oracle_detect` (backbone -> BEV -> 2-D backbone -> head maps) in FLOAT64 on the same fp32 weights and the
    same fp32 voxel features: the yardstick of the error-budget test (an fp32 evaluation's own rounding noise is what the HIP
    engines are allowed; against float64 it can be measured for each of them separately)."""
    from oracle import dense, sparse as osp, voxelize as ov
    if dynamic:
        pb = np.concatenate([np.zeros((points.shape[0], 1), np.float32), points], 1)
        feats, coords = ov.dynamic_mean_vfe(pb, info.point_cloud_range, info.voxel_size)
    else:
        vox, czyx, nump = ov.hard_voxelize(points, info.point_cloud_range, info.voxel_size, 5, info.max_voxels['test'])
        feats = ov.mean_vfe(vox, nump)
        coords = np.concatenate([np.zeros((czyx.shape[0], 1), np.int32), czyx], 1)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    grid = info.grid_size
    res = osp.backbone_forward(sd64, feats, coords, [int(grid[2]) + 1, int(grid[1]), int(grid[0])], dtype=torch.float64)
    x, oc, shape = res['encoded']
    bev = osp.to_bev(x, oc, shape, 1)
    f2d = dense.bev_backbone_forward(sd64, bev)
    pred = dense.center_head_forward(sd64, f2d)
    return {'backbone': res, 'bev': bev, 'f2d': f2d, 'pred': pred}


def match_boxes(a_boxes, a_scores, b_boxes, b_scores, tol=1e-3):
    """Greedy one-to-one match by centre distance; returns (n_matched, max_abs_diff over matched)."""
    a_boxes, b_boxes = np.asarray(a_boxes), np.asarray(b_boxes)
    if a_boxes.shape[0] == 0 or b_boxes.shape[0] == 0:
        return 0, 0.0
    d = np.abs(a_boxes[:, None, :3] - b_boxes[None, :, :3]).max(-1)
    used = set()
    n, worst = 0, 0.0
    for i in range(a_boxes.shape[0]):
        j = int(np.argmin(d[i]))
        if d[i, j] <= tol and j not in used:
            diff = np.abs(a_boxes[i] - b_boxes[j])
            diff[6] = min(diff[6], abs(diff[6] - 2 * np.pi))
            sd = abs(float(a_scores[i]) - float(b_scores[j]))
            if diff.max() <= tol and sd <= tol:
                used.add(j)
                n += 1
                worst = max(worst, float(diff.max()), sd)
    return n, worst


def canon_order(coords, shape):
    """Permutation that puts rows given by coords (m,4) [b,z,y,x] (any storage order, e.g. the brick order of the backbone's levels)
    into the canonical order of the parity statements: ascending linear key ((b*D+z)*H+y)*W+x."""
    c = np.asarray(coords).astype(np.int64)
    d, h, w = (int(x) for x in shape)
    return np.argsort(((c[:, 0] * d + c[:, 1]) * h + c[:, 2]) * w + c[:, 3], kind='stable')


def canon_table(nbr, out_order, in_order):
    """Neighbour table (kvol, m_out) in STORAGE rows -> the same table in canonical rows on both sides."""
    nbr = np.asarray(nbr)
    inv_in = np.empty(in_order.size, np.int64)
    inv_in[in_order] = np.arange(in_order.size)
    t = nbr[:, out_order]
    return np.where(t >= 0, inv_in[np.maximum(t, 0)], -1).astype(np.int32)


def canon_tensor(t):
    """(indices, features) of a SparseConvTensor sorted into the canonical row order."""
    idx = t.indices.cpu().numpy()
    o = canon_order(idx, t.spatial_shape)
    return idx[o], t.features.cpu()[torch.from_numpy(o)]
