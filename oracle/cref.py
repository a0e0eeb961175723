"""ORACLE (test infrastructure only): ctypes view of oracle/c/oracle.c, built with gcc on demand."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liboracle.so')
_SRC = os.path.join(_HERE, 'c', 'oracle.c')


def build(force=False):
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-fno-fast-math', '-shared', '-fPIC',
                               '-o', _SO, _SRC, '-lm'])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.orc_box_overlap.restype = ctypes.c_float
        _lib.orc_iou_bev.restype = ctypes.c_float
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def voxelize_hard(points, pc_range, voxel_size, max_points, max_voxels):
    """Sequential definition (oracle.c:orc_voxelize_hard). points (N,C) float32."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    n, c = points.shape
    rng = np.ascontiguousarray(pc_range, dtype=np.float32)
    vs = np.ascontiguousarray(voxel_size, dtype=np.float32)
    grid = np.round((rng[3:6].astype(np.float64) - rng[0:3]) / np.asarray(voxel_size, np.float64)).astype(np.int32)
    lut = np.full(int(grid[0]) * int(grid[1]) * int(grid[2]), -1, dtype=np.int32)
    voxels = np.zeros((max_voxels, max_points, c), dtype=np.float32)
    coords = np.zeros((max_voxels, 3), dtype=np.int32)
    num = np.zeros((max_voxels,), dtype=np.int32)
    m = lib().orc_voxelize_hard(_p(points, ctypes.c_float), n, c, _p(rng, ctypes.c_float),
                                _p(vs, ctypes.c_float), _p(grid, ctypes.c_int), max_points, max_voxels,
                                _p(voxels, ctypes.c_float), _p(coords, ctypes.c_int),
                                _p(num, ctypes.c_int), _p(lut, ctypes.c_int))
    return voxels[:m].copy(), coords[:m].copy(), num[:m].copy()


def boxes_overlap_bev(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    lib().orc_boxes_overlap_bev(_p(a, ctypes.c_float), a.shape[0], _p(b, ctypes.c_float), b.shape[0],
                                _p(out, ctypes.c_float))
    return out


def boxes_iou_bev(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    lib().orc_boxes_iou_bev(_p(a, ctypes.c_float), a.shape[0], _p(b, ctypes.c_float), b.shape[0],
                            _p(out, ctypes.c_float))
    return out


def nms_sorted(boxes_sorted, thr):
    """boxes already in descending-score order; returns kept indices (int64)."""
    b = np.ascontiguousarray(boxes_sorted[:, :7], np.float32)
    keep = np.zeros((b.shape[0],), np.int64)
    nk = lib().orc_nms(_p(b, ctypes.c_float), b.shape[0], ctypes.c_float(thr), _p(keep, ctypes.c_longlong))
    return keep[:nk].copy()


def points_in_boxes_margin(points_xyz, boxes, margin):
    p = np.ascontiguousarray(points_xyz[:, :3], np.float32)
    b = np.ascontiguousarray(boxes[:, :7], np.float32)
    mask = np.zeros((b.shape[0], p.shape[0]), np.int32)
    lib().orc_points_in_boxes_margin(_p(b, ctypes.c_float), b.shape[0], _p(p, ctypes.c_float), p.shape[0], ctypes.c_float(margin),
                                     _p(mask, ctypes.c_int))
    return mask


def points_in_boxes_v2(points_xyz, boxes):
    p = np.ascontiguousarray(points_xyz[:, :3], np.float32)
    b = np.ascontiguousarray(boxes[:, :7], np.float32)
    mask = np.zeros((b.shape[0], p.shape[0]), np.int32)
    lib().orc_points_in_boxes_v2(_p(b, ctypes.c_float), b.shape[0], _p(p, ctypes.c_float), p.shape[0],
                                 _p(mask, ctypes.c_int))
    return mask
