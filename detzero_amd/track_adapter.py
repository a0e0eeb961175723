"""Tracker input adapter: from the gathered per-frame boxes to what DetZero's CPU tracker consumes.

Mirror of the reference's tracking-side pre-processing:
  * ``tracking/detzero_track/utils/data_utils.py:8-30``      frame_list_to_dict / sequence_list_to_dict / dict_to_sequence_list
  * ``tracking/detzero_track/utils/transform_utils.py:4-59`` yaw_filter / get_inverse_transform_mat / transform_boxes3d
  * ``tracking/detzero_track/datasets/data_processor.py:14-163``  DataProcessor: a config-named queue of
    heading_process, points_in_box, low_confidence_box_filter, transform_to_global, overlap_box_filter
    (``tracking/tools/cfgs/tk_dataset_cfgs/waymo_dataset.yaml:7-22``).
Same names, same dict keys, same quirks (listed at each function), so ``WaymoTrackDataset`` can take this
DataProcessor unchanged.  The one device computation of this stage - the all-pairs rotated BEV overlap matrix of a
frame's detections (``bev_overlap_gpu``, data_association/distance.py:44-64) and the point counts per box - runs on
the HIP kernels (dz_boxes_overlap_bev, dz_points_in_boxes_v2) through ``iou3d_nms_utils`` / ``roiaware_pool3d_utils``;
the greedy, order-dependent selection that follows is a few hundred scalar steps per frame and stays on the host,
exactly as in the reference.  ``overlap_fn`` / ``count_fn`` can be injected (the CPU tests pass the oracle's).
"""
import copy
import os
from functools import partial

import numpy as np

TWO_PI = 2.0 * np.pi


# ------------------------------------------------------------------------------------------------
# containers (data_utils.py:8-30)
# ------------------------------------------------------------------------------------------------
def frame_list_to_dict(data):
    """list of frame dicts -> {str(sample_idx): frame}"""
    return {str(item['sample_idx']): item for item in data}


def sequence_list_to_dict(data):
    """list of frame dicts -> {sequence_name: {str(sample_idx or frame_id): frame}} (insertion order kept)"""
    out = {}
    for item in data:
        key = str(item['sample_idx']) if 'sample_idx' in item else str(item['frame_id'])
        out.setdefault(item['sequence_name'], {})[key] = item
    return out


def dict_to_sequence_list(data):
    return [frame for seq in data.values() for frame in seq.values()]


# ------------------------------------------------------------------------------------------------
# geometry (transform_utils.py)
# ------------------------------------------------------------------------------------------------
def yaw_filter(yaw):
    """Heading into (-pi, pi].  Arrays are wrapped IN PLACE; a scalar is only touched when |yaw| >= 2*pi
    (transform_utils.py:14-24 - a scalar in (pi, 2*pi) is returned as it came)."""
    if isinstance(yaw, np.ndarray):
        big = np.abs(yaw) >= TWO_PI
        yaw[big] = yaw[big] - np.floor(yaw[big] / TWO_PI) * TWO_PI
        yaw[yaw > np.pi] -= TWO_PI
        yaw[yaw <= -np.pi] += TWO_PI
        return yaw
    if np.abs(yaw) >= TWO_PI:
        yaw = yaw - np.floor(yaw / TWO_PI) * TWO_PI
        if yaw > np.pi:
            yaw -= TWO_PI
        if yaw <= -np.pi:
            yaw += TWO_PI
    return yaw


def get_inverse_transform_mat(src_pose):
    """Inverse of a rigid 4x4 pose, float32 (transform_utils.py:29-41)."""
    rot_t = src_pose[:3, :3].T
    inv = np.zeros((4, 4), dtype=np.float32)
    inv[:3, :3] = rot_t
    inv[:3, 3:] = -(rot_t @ src_pose[:3, 3:])
    inv[3, 3] = 1
    return inv


def transform_boxes3d(boxes, pose, inverse=False):
    """(N,7) boxes through a 4x4 pose: centres as homogeneous rows times pose^T, heading + atan2(r10, r00) wrapped,
    sizes untouched (transform_utils.py:43-59)."""
    if inverse:
        pose = get_inverse_transform_mat(pose)
    homo = np.concatenate([boxes[:, :3], np.ones((boxes.shape[0], 1))], axis=-1) @ pose.T
    heading = yaw_filter(boxes[:, [6]] + np.arctan2(pose[1, 0], pose[0, 0]))
    return np.concatenate([homo[:, :3], boxes[:, 3:6], heading], axis=-1)


# ------------------------------------------------------------------------------------------------
# device helpers
# ------------------------------------------------------------------------------------------------
def bev_overlap_gpu(boxes_a, boxes_b):
    """(N,7),(M,7) numpy or device tensors -> (N,M) float32 numpy overlap areas (distance.py:44-64) on the HIP kernel."""
    import torch
    from . import iou3d_nms_utils
    n, m = boxes_a.shape[0], boxes_b.shape[0]
    if n == 0 or m == 0:
        return np.zeros((n, m), dtype=np.float32)
    dev = torch.device('cuda', torch.cuda.current_device())
    a = boxes_a if torch.is_tensor(boxes_a) else torch.from_numpy(np.ascontiguousarray(boxes_a[:, :7], dtype=np.float32))
    b = boxes_b if torch.is_tensor(boxes_b) else torch.from_numpy(np.ascontiguousarray(boxes_b[:, :7], dtype=np.float32))
    return iou3d_nms_utils.boxes_overlap_bev_gpu(a.to(dev), b.to(dev)).cpu().numpy()


def points_in_boxes_num_gpu(points_xyz, boxes):
    """Number of points inside each box (roiaware_pool3d_utils.points_in_boxes_num_gpu as used at
    data_processor.py:64-69) -> (T,) int32 numpy: dz_points_in_boxes_count (per-wavefront popcounts + atomics into a (T,)
    counter; the dense (T, M) mask of points_in_boxes_gpu_v2 is never built)."""
    import torch
    from . import lib as L
    t = boxes.shape[0]
    if t == 0:
        return np.zeros((0,), dtype=np.int32)
    dev = points_xyz.device if torch.is_tensor(points_xyz) else torch.device('cuda', torch.cuda.current_device())
    p = points_xyz if torch.is_tensor(points_xyz) else torch.from_numpy(np.ascontiguousarray(points_xyz[:, :3], dtype=np.float32))
    b = boxes if torch.is_tensor(boxes) else torch.from_numpy(np.ascontiguousarray(boxes[:, :7], dtype=np.float32))
    p = p[:, :3].to(dev, torch.float32).contiguous()
    b = b[:, :7].to(dev, torch.float32).contiguous()
    counts = torch.empty((t,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = L.load().dz_points_in_boxes_count(L.ptr(b), L.ptr(p), t, p.shape[0], L.ptr(counts), L.stream())
    L.check(rc, 'dz_points_in_boxes_count')
    return counts.cpu().numpy()


# ------------------------------------------------------------------------------------------------
# DataProcessor (data_processor.py:14-163)
# ------------------------------------------------------------------------------------------------
class DataProcessor(object):
    """``processor_configs``: list of objects with ``.NAME`` (+ per-processor fields), applied in order to every
    frame of ``forward``'s ``{sample_idx: frame_dict}`` in ascending integer sample order.  Returns
    ``(processed, removed)``: a processor that returns a tuple contributes its second element to ``removed``."""

    def __init__(self, processor_configs, lidar_path=None, overlap_fn=None, count_fn=None):
        self.lidar_path = lidar_path
        self.ignore_key_list = ['sequence_name', 'timestamp', 'pose', 'frame_id']
        self.overlap_fn = overlap_fn if overlap_fn is not None else bev_overlap_gpu
        self.count_fn = count_fn if count_fn is not None else points_in_boxes_num_gpu
        self.data_processor_queue = [getattr(self, cfg.NAME)(config=cfg) for cfg in processor_configs]

    def forward(self, data_dict):
        processed, removed = {}, {}
        for sample_idx in sorted(data_dict.keys(), key=int):
            cur = data_dict[sample_idx]
            for proc in self.data_processor_queue:
                cur = proc(data_dict=cur)
                if isinstance(cur, tuple):
                    cur, removed[sample_idx] = cur
            processed[sample_idx] = cur
        return processed, removed

    # ---- processors -----------------------------------------------------------------------------
    def heading_process(self, data_dict=None, config=None):
        if data_dict is None:
            return partial(self.heading_process, config=config)
        if data_dict.get('boxes_lidar', None) is not None:
            data_dict['boxes_lidar'][:, 6] = yaw_filter(data_dict['boxes_lidar'][:, 6])
        return data_dict

    def points_in_box(self, data_dict=None, config=None):
        """Loads the frame's point cloud ``<lidar_path>/segment-<seq>/<frame_id:04d>.npy`` and counts points per box."""
        if data_dict is None:
            return partial(self.points_in_box, config=config)
        if data_dict.get('boxes_lidar', None) is not None:
            seq = data_dict['sequence_name']
            seq = seq if 'segment-' in seq else 'segment-' + seq
            fname = ('0000' + str(data_dict['frame_id']))[-4:] + '.npy'
            points = np.load(os.path.join(self.lidar_path, seq, fname))
            data_dict['num_points'] = self.count_fn(points[:, :3], data_dict['boxes_lidar'][:, :7])
        return data_dict

    def low_confidence_box_filter(self, data_dict=None, config=None, threshold=0.):
        if data_dict is None:
            return partial(self.low_confidence_box_filter, config=config)
        if data_dict.get('score', None) is not None:
            keep = data_dict['score'] >= config.THRESHOLD
            for key in list(data_dict.keys()):
                if key not in self.ignore_key_list:
                    data_dict[key] = data_dict[key][keep]
        return data_dict

    def transform_to_global(self, data_dict=None, config=None):
        if data_dict is None:
            return partial(self.transform_to_global, config=config)
        if data_dict.get('pose', None) is not None:
            data_dict['boxes_global'] = transform_boxes3d(data_dict['boxes_lidar'], data_dict['pose'])
        return data_dict

    def overlap_box_filter(self, data_dict=None, config=None):
        """Greedy de-duplication of a frame's detections on BEV overlap (data_processor.py:102-163).

        For every box i in input order that has not itself been chosen yet: the group = all boxes j whose overlap
        with i covers at least CLASS_THRESHOLD[name_i] of box i's footprint (dx*dy of box i, read AFTER any earlier
        in-place merge); the group's highest-scoring member is kept.  METHOD 'weigthed_size' [sic] / 'merge_box'
        additionally overwrite the kept box's size / centre+size with the score-weighted mean of the group.
        Quirks kept: boxes that were only *members* of an earlier group still seed their own group later; a frame
        without detections is returned as a bare dict instead of a (dict, removed) tuple."""
        if data_dict is None:
            return partial(self.overlap_box_filter, config=config)
        removed = {}
        if data_dict.get('boxes_lidar', None) is not None:
            boxes, names, scores = data_dict['boxes_lidar'], data_dict['name'], data_dict['score']
            if len(names) == 0:
                return data_dict
            overlap = self.overlap_fn(boxes[:, :7], boxes[:, :7])
            n = len(boxes)
            chosen = np.zeros(n, dtype=bool)
            for i in range(n):
                if chosen[i]:
                    continue
                rate = overlap[i] / (boxes[i, 3] * boxes[i, 4])
                group = np.flatnonzero(rate >= config.CLASS_THRESHOLD[names[i]])
                g_score = scores[group]
                best = group[np.argsort(g_score)[-1]]
                chosen[best] = True
                if config.METHOD in ('weigthed_size', 'merge_box'):
                    cols = slice(3, 6) if config.METHOD == 'weigthed_size' else slice(0, 6)
                    width = cols.stop - cols.start
                    wts = np.repeat(g_score.reshape(-1, 1), width, axis=1)
                    data_dict['boxes_lidar'][best, cols] = np.sum(boxes[group][:, cols] * wts, axis=0) / (np.sum(g_score) + 1e-9)
            keep_idx = np.flatnonzero(chosen)
            drop_idx = np.flatnonzero(~chosen)
            for key in data_dict.keys():
                if key in self.ignore_key_list:
                    removed[key] = copy.deepcopy(data_dict[key])
                else:
                    removed[key] = copy.deepcopy(data_dict[key][drop_idx])
                    data_dict[key] = data_dict[key][keep_idx]
        return data_dict, removed


def prepare_tracker_input(annos, processor_configs, lidar_path=None, overlap_fn=None, count_fn=None):
    """Detection result records (what ``frame_parallel.run_frame_parallel`` / result.pkl hold: name, score, boxes_lidar,
    sequence_name, frame_id, pose, ...) -> {sequence_name: (processed frames, removed boxes)} ready for the tracker
    (WaymoTrackDataset.__getitem__, tracking/detzero_track/datasets/waymo/waymo_dataset.py)."""
    proc = DataProcessor(processor_configs, lidar_path, overlap_fn, count_fn)
    return {seq: proc.forward(frames) for seq, frames in sequence_list_to_dict(annos).items()}
