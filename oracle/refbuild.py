"""ORACLE (test infrastructure only): build and call ``oracle/_ref`` - the reference's OWN rotated
BEV IoU (utils/detzero_utils/ops/iou3d_nms/src/iou3d_cpu.cpp) compiled with g++ from the sources
where they lie under /root/reference.  Only possible in the build container (the GPU box has no
/root/reference; the prebuilt .so travels with the snapshot).  Outputs go to oracle/_ref/ (git-ignored).
"""
import ctypes
import os
import subprocess
import sysconfig

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC_DIR = '/root/reference/utils/detzero_utils/ops/iou3d_nms/src'
_SO = os.path.join(_HERE, '_ref', 'libiou3d_cpu_ref.so')


def available():
    return os.path.exists(_SO) or os.path.isdir(REF_SRC_DIR)


def build(force=False):
    if os.path.exists(_SO) and not force:
        return _SO
    if not os.path.isdir(REF_SRC_DIR):
        raise RuntimeError('reference sources not present and %s not prebuilt' % _SO)
    import torch
    from torch.utils import cpp_extension
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    inc = ['-I' + p for p in cpp_extension.include_paths()]
    inc += ['-I' + sysconfig.get_paths()['include'], '-I' + os.path.join(_HERE, 'ref_build', 'stubs'), '-I' + REF_SRC_DIR]
    libdir = os.path.join(os.path.dirname(torch.__file__), 'lib')
    cmd = ['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-ffp-contract=off', '-w',
           '-D_GLIBCXX_USE_CXX11_ABI=%d' % int(torch._C._GLIBCXX_USE_CXX11_ABI)] + inc + \
          [os.path.join(_HERE, 'ref_build', 'iou3d_cpu_wrap.cpp'), '-o', _SO, '-L' + libdir, '-Wl,-rpath,' + libdir,
           '-ltorch', '-ltorch_cpu', '-lc10']
    subprocess.check_call(cmd)
    return _SO


ROI_SRC_DIR = '/root/reference/utils/detzero_utils/ops/roiaware_pool3d/src'
_ROI_SO = os.path.join(_HERE, '_ref', 'libroiaware_cpu_ref.so')


def build_roiaware(force=False):
    """The reference's roiaware_pool3d.cpp (host part: points_in_boxes_cpu) compiled the same way."""
    if os.path.exists(_ROI_SO) and not force:
        return _ROI_SO
    if not os.path.isdir(ROI_SRC_DIR):
        raise RuntimeError('reference sources not present and %s not prebuilt' % _ROI_SO)
    import torch
    from torch.utils import cpp_extension
    os.makedirs(os.path.dirname(_ROI_SO), exist_ok=True)
    inc = ['-I' + p for p in cpp_extension.include_paths()]
    inc += ['-I' + sysconfig.get_paths()['include'], '-I' + os.path.join(_HERE, 'ref_build', 'stubs'), '-I' + ROI_SRC_DIR]
    libdir = os.path.join(os.path.dirname(torch.__file__), 'lib')
    cmd = ['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-ffp-contract=off', '-w',
           '-D_GLIBCXX_USE_CXX11_ABI=%d' % int(torch._C._GLIBCXX_USE_CXX11_ABI)] + inc + \
          [os.path.join(_HERE, 'ref_build', 'roiaware_cpu_wrap.cpp'), '-o', _ROI_SO, '-L' + libdir, '-Wl,-rpath,' + libdir,
           '-ltorch', '-ltorch_cpu', '-lc10', '-ltorch_python']
    subprocess.check_call(cmd)
    return _ROI_SO


_roi_lib = None


def points_in_boxes_cpu_reference(points_xyz, boxes):
    """(T,7) boxes, (M,3) points -> (T,M) int32 flags from the reference's points_in_boxes_cpu."""
    global _roi_lib
    if _roi_lib is None:
        import torch  # noqa: F401
        _roi_lib = ctypes.CDLL(build_roiaware())
    p = np.ascontiguousarray(points_xyz[:, :3], np.float32)
    b = np.ascontiguousarray(boxes[:, :7], np.float32)
    out = np.zeros((b.shape[0], p.shape[0]), np.int32)
    fp = ctypes.POINTER(ctypes.c_float)
    _roi_lib.ref_points_in_boxes_cpu(b.ctypes.data_as(fp), b.shape[0], p.ctypes.data_as(fp), p.shape[0],
                                     out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return out


_lib = None


def _load():
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (libtorch must be loaded first)
        _lib = ctypes.CDLL(build())
    return _lib


def boxes_iou_bev_reference(a, b):
    a = np.ascontiguousarray(a[:, :7], np.float32)
    b = np.ascontiguousarray(b[:, :7], np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    _load().ref_boxes_iou_bev_cpu(a.ctypes.data_as(fp), a.shape[0], b.ctypes.data_as(fp), b.shape[0],
                                  out.ctypes.data_as(fp))
    return out


def nms_with_reference_iou(boxes_sorted, thresh):
    """Sweep of iou3d_nms.cpp:145-156 driven by the reference's own IoU."""
    n = boxes_sorted.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64)
    iou = boxes_iou_bev_reference(boxes_sorted, boxes_sorted)
    removed = np.zeros(n, bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        removed[i + 1:] |= iou[i, i + 1:] > thresh
    return np.asarray(keep, np.int64)
