"""ORACLE (test infrastructure only): the three CUDA kernels under the PDV second stage, restated in numpy.

The reference's PDV head (detection/detzero_det/models/centerpoint_modules/pdv_head.py) is Python on top of three compiled
CUDA extensions that cannot be built here (CUDAExtension, <cuda.h>).  Their kernels are short and read literally:
  * ball_query_count_kernel_stack   utils/detzero_utils/ops/pointnet2/pointnet2_stack/src/ball_query_count_gpu.cu:16-62
  * group_points_kernel_stack       .../src/group_points_gpu.cu:71-102
  * points_in_multi_boxes_kernel    utils/detzero_utils/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:16-36,377-404
tests/golden/gen_pdv_golden.py installs these functions in place of the extension modules and then runs the reference's OWN
Python classes (PDVHead, StackSAModuleMSGAttention, QueryAndGroup, the KDE, TransformerEncoder, density / voxel aggregation
utilities) on the CPU; everything above the three kernels in the fixture is therefore the reference itself.
All arithmetic float32, one rounding per operation (nvcc may contract a*a + b*b into an FMA; a point exactly on a ball or
box boundary could differ - the synthetic inputs keep clear of that, see the margin checks in the generator).
"""
import numpy as np


def ball_query_count(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
    """-> idx (M, nsample) int32, -1 filled: for every query the first `nsample` points of ITS batch item, in index order,
    with squared distance < radius^2; indices are relative to the batch item's first point; a ball without any point has
    idx[0] = -1 (all of its row stays -1)."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    new_xyz = np.ascontiguousarray(new_xyz, np.float32)
    m = new_xyz.shape[0]
    idx = np.full((m, nsample), -1, np.int32)
    r2 = np.float32(radius) * np.float32(radius)
    p_start = np.concatenate([[0], np.cumsum(xyz_batch_cnt)]).astype(np.int64)
    q_start = np.concatenate([[0], np.cumsum(new_xyz_batch_cnt)]).astype(np.int64)
    for b in range(len(xyz_batch_cnt)):
        pts = xyz[p_start[b]:p_start[b + 1]]
        for q in range(q_start[b], q_start[b + 1]):
            d = new_xyz[q][None, :] - pts
            sq = (d * d).astype(np.float32)
            d2 = ((sq[:, 0] + sq[:, 1]).astype(np.float32) + sq[:, 2]).astype(np.float32)
            hit = np.nonzero(d2 < r2)[0][:nsample]
            idx[q, :hit.size] = hit
    return idx


def group_points(features, features_batch_cnt, idx, idx_batch_cnt):
    """-> (M, C, nsample): features[batch start + idx[m, s], c]."""
    features = np.ascontiguousarray(features, np.float32)
    m, ns = idx.shape
    out = np.zeros((m, features.shape[1], ns), np.float32)
    p_start = np.concatenate([[0], np.cumsum(features_batch_cnt)]).astype(np.int64)
    q_start = np.concatenate([[0], np.cumsum(idx_batch_cnt)]).astype(np.int64)
    for b in range(len(idx_batch_cnt)):
        rows = idx[q_start[b]:q_start[b + 1]].astype(np.int64) + p_start[b]
        out[q_start[b]:q_start[b + 1]] = np.transpose(features[rows], (0, 2, 1))
    return out


def point_in_box(pts, box):
    """check_pt_in_box3d: |z - cz| <= dz/2, then the rotated xy test with margin 1e-5 (float32 cos / sin of -heading)."""
    pts = pts.astype(np.float32)
    cx, cy, cz, dx, dy, dz, rz = (np.float32(v) for v in box[:7])
    zok = ~(np.abs(pts[:, 2] - cz).astype(np.float64) > np.float64(dz) / 2.0)
    cosa, sina = np.float32(np.cos(np.float32(-rz))), np.float32(np.sin(np.float32(-rz)))
    sx, sy = pts[:, 0] - cx, pts[:, 1] - cy
    lx = (sx * cosa).astype(np.float32) + (sy * (-sina)).astype(np.float32)
    ly = (sx * sina).astype(np.float32) + (sy * cosa).astype(np.float32)
    inx = np.abs(lx).astype(np.float64) < np.float64(dx) / 2.0 + np.float64(np.float32(1e-5))
    iny = np.abs(ly).astype(np.float64) < np.float64(dy) / 2.0 + np.float64(np.float32(1e-5))
    return zok & inx & iny


def points_in_multi_boxes(points, boxes, max_num_boxes):
    """points (B, M, 3), boxes (B, T, 7) -> (B, M, max_num_boxes) int32: per point the first max_num_boxes boxes (in box order)
    that contain it, -1 filled."""
    bsz, m, _ = points.shape
    out = np.full((bsz, m, max_num_boxes), -1, np.int32)
    for b in range(bsz):
        fill = np.zeros(m, np.int64)
        for k in range(boxes.shape[1]):
            inside = point_in_box(points[b], boxes[b, k]) & (fill < max_num_boxes)
            sel = np.nonzero(inside)[0]
            out[b, sel, fill[sel]] = k
            fill[sel] += 1
    return out
