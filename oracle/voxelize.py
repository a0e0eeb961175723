"""ORACLE (test infrastructure only): points -> voxels, numpy restatement.

Reference (paths under /root/reference/detection/detzero_det):
  * range mask       - utils common_utils.mask_points_by_range (utils/detzero_utils/common_utils.py:247-250)
  * hard voxelizer   - datasets/processor/data_processor.py:61-91 -> spconv Point2VoxelCPU3d
                       (un-vendored; semantics per SURVEY.md App. C; sequential definition lives in
                       oracle/c/oracle.c:orc_voxelize_hard, this file is the vectorised equivalent)
  * MeanVFE          - models/centerpoint_modules/vfe.py:66-83
  * DynamicMeanVFE   - models/centerpoint_modules/vfe.py:109-147
"""
import numpy as np


def grid_size_of(pc_range, voxel_size):
    """data_processor.py:63-65 (np.round of the float64 quotient)."""
    pc_range = np.asarray(pc_range, dtype=np.float32)
    g = (pc_range[3:6] - pc_range[0:3]) / np.array(voxel_size)
    return np.round(g).astype(np.int64)


def mask_points_by_range(points, limit_range):
    """xy only, bounds inclusive (common_utils.py:247-250)."""
    return (points[:, 0] >= limit_range[0]) & (points[:, 0] <= limit_range[3]) & \
           (points[:, 1] >= limit_range[1]) & (points[:, 1] <= limit_range[4])


def point_voxel_coords(xyz, pc_range, voxel_size):
    """c_j = floor((p_j - lo_j) / vs_j), everything IEEE fp32 (vfe.py:124; spconv CPU loop).
    Returns int32 (N,3) in x,y,z order and the in-grid mask."""
    lo = np.asarray(pc_range, dtype=np.float32)[0:3]
    vs = np.asarray(voxel_size, dtype=np.float32)
    grid = grid_size_of(pc_range, voxel_size)
    d = xyz.astype(np.float32) - lo[None, :]
    q = (d / vs[None, :]).astype(np.float32)
    c = np.floor(q).astype(np.int32)
    ok = np.all((c >= 0) & (c < grid[None, :].astype(np.int32)), axis=1)
    return c, ok


def hard_voxelize(points, pc_range, voxel_size, max_points, max_voxels):
    """Vectorised equivalent of the sequential definition.  Returns
    voxels (M,max_points,C) f32, coords (M,3) int32 zyx, num_points (M,) int32, in first-appearance order."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    n, c = points.shape
    grid = grid_size_of(pc_range, voxel_size)
    cxyz, ok = point_voxel_coords(points[:, :3], pc_range, voxel_size)
    idx = np.nonzero(ok)[0]
    cc = cxyz[idx].astype(np.int64)
    key = (cc[:, 2] * grid[1] + cc[:, 1]) * grid[0] + cc[:, 0]
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    order = np.argsort(first, kind='stable')            # uniq index sorted by first appearance
    vid_of_uniq = np.empty_like(order)
    vid_of_uniq[order] = np.arange(order.size)
    vid = vid_of_uniq[inv]                               # voxel id per valid point
    keep_pt = vid < max_voxels
    idx, vid, cc = idx[keep_pt], vid[keep_pt], cc[keep_pt]
    m = int(min(order.size, max_voxels))
    # rank of each point inside its voxel (input order)
    srt = np.argsort(vid, kind='stable')
    vs_sorted = vid[srt]
    starts = np.searchsorted(vs_sorted, np.arange(m), side='left')
    rank_sorted = np.arange(vs_sorted.size) - starts[vs_sorted]
    rank = np.empty_like(rank_sorted)
    rank[srt] = rank_sorted
    sel = rank < max_points
    voxels = np.zeros((m, max_points, c), dtype=np.float32)
    voxels[vid[sel], rank[sel]] = points[idx[sel]]
    counts = np.bincount(vid, minlength=m)
    num_points = np.minimum(counts, max_points).astype(np.int32)
    coords = np.zeros((m, 3), dtype=np.int32)
    firsts = first[order][:m]
    coords[:, 0] = cc_full(cxyz, np.nonzero(ok)[0], firsts, 2)
    coords[:, 1] = cc_full(cxyz, np.nonzero(ok)[0], firsts, 1)
    coords[:, 2] = cc_full(cxyz, np.nonzero(ok)[0], firsts, 0)
    return voxels, coords, num_points


def cc_full(cxyz, valid_idx, firsts, axis):
    return cxyz[valid_idx[firsts], axis]


def mean_vfe(voxels, num_points):
    """vfe.py:76-80: sum over all slots / clamp_min(num_points, 1), fp32."""
    s = voxels.astype(np.float32).sum(axis=1, dtype=np.float32)
    norm = np.maximum(num_points.astype(np.float32), 1.0)[:, None]
    return (s / norm).astype(np.float32)


def dynamic_mean_vfe(points_b, pc_range, voxel_size):
    """vfe.py:109-147.  points_b (N, 1+C) ``[b, x, y, z, ...]`` float32.
    Returns features (M,C) f32 (fp64 accumulation, rounded once) and coords (M,4) int32
    ``[b,z,y,x]`` in ascending merge-key order (x-major key, vfe.py:128-131)."""
    points_b = np.ascontiguousarray(points_b, dtype=np.float32)
    grid = grid_size_of(pc_range, voxel_size)
    cxyz, ok = point_voxel_coords(points_b[:, 1:4], pc_range, voxel_size)
    pts = points_b[ok]
    cc = cxyz[ok].astype(np.int64)
    sxyz, syz, sz = int(grid[0] * grid[1] * grid[2]), int(grid[1] * grid[2]), int(grid[2])
    key = pts[:, 0].astype(np.int64) * sxyz + cc[:, 0] * syz + cc[:, 1] * sz + cc[:, 2]
    uniq, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    data = pts[:, 1:].astype(np.float64)
    sums = np.zeros((uniq.size, data.shape[1]), dtype=np.float64)
    np.add.at(sums, inv, data)
    feats = (sums / cnt[:, None]).astype(np.float32)
    b = uniq // sxyz
    x = (uniq % sxyz) // syz
    y = (uniq % syz) // sz
    z = uniq % sz
    coords = np.stack([b, z, y, x], axis=1).astype(np.int32)
    return feats, coords
