"""DataProcessor under the reference's config names, voxelizing on the MI355X.

Mirror of /root/reference/detection/detzero_det/datasets/processor/data_processor.py:11-138:
name-keyed processor queue (``getattr(self, cfg.NAME)``), each entry called once with
``data_dict=None`` at construction (returns a partial, sets grid_size / voxel_size) and then per
sample.  ``transform_points_to_voxels`` replaces spconv's CPU ``Point2VoxelCPU3d`` with
``dz_voxelize_hard``; because forked DataLoader workers cannot own a HIP context it must run in the
process that owns the GPU (SURVEY.md §7 hard part 7) - points may be numpy (copied H2D here) or
already-resident device tensors (kept on the device, zero copies).
"""
from functools import partial

import numpy as np
import torch

from . import ops


def mask_points_by_range(points, limit_range):
    """utils/detzero_utils/common_utils.py:247-250 - xy only, bounds inclusive."""
    return (points[:, 0] >= limit_range[0]) & (points[:, 0] <= limit_range[3]) & \
           (points[:, 1] >= limit_range[1]) & (points[:, 1] <= limit_range[4])


class DataProcessor(object):
    def __init__(self, processor_configs, point_cloud_range, training, num_point_features):
        self.point_cloud_range = np.asarray(point_cloud_range, dtype=np.float32)
        self.training = training
        self.num_point_features = num_point_features
        self.mode = 'train' if training else 'test'
        self.grid_size = self.voxel_size = None
        self.data_processor_queue = []
        for cur_cfg in processor_configs:
            cur_processor = getattr(self, cur_cfg.NAME)(config=cur_cfg)
            self.data_processor_queue.append(cur_processor)

    def mask_points_and_boxes_outside_range(self, data_dict=None, config=None):
        if data_dict is None:
            return partial(self.mask_points_and_boxes_outside_range, config=config)
        pts = data_dict['points']
        mask = mask_points_by_range(pts, self.point_cloud_range)
        data_dict['points'] = pts[mask]
        return data_dict

    def shuffle_points(self, data_dict=None, config=None):
        if data_dict is None:
            return partial(self.shuffle_points, config=config)
        if config.SHUFFLE_ENABLED[self.mode]:
            pts = data_dict['points']
            idx = np.random.permutation(pts.shape[0])
            data_dict['points'] = pts[torch.from_numpy(idx).to(pts.device)] if torch.is_tensor(pts) else pts[idx]
        return data_dict

    def _set_grid(self, config):
        grid_size = (self.point_cloud_range[3:6] - self.point_cloud_range[0:3]) / np.array(config.VOXEL_SIZE)
        self.grid_size = np.round(grid_size).astype(np.int64)
        self.voxel_size = config.VOXEL_SIZE

    def transform_points_to_voxels_placeholder(self, data_dict=None, config=None):
        if data_dict is None:
            self._set_grid(config)
            return partial(self.transform_points_to_voxels_placeholder, config=config)
        return data_dict

    def transform_points_to_voxels(self, data_dict=None, config=None):
        if data_dict is None:
            self._set_grid(config)
            return partial(self.transform_points_to_voxels, config=config)
        points = data_dict['points']
        as_numpy = not torch.is_tensor(points)
        if as_numpy:
            points = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).cuda()
        voxels, coords, num_points = ops.voxelize_hard(
            points.float().contiguous(), self.point_cloud_range, config.VOXEL_SIZE, config.MAX_POINTS_PER_VOXEL,
            config.MAX_NUMBER_OF_VOXELS[self.mode])
        if not data_dict.get('use_lead_xyz', True):
            voxels = voxels[..., 3:]
        if as_numpy:
            voxels, coords, num_points = voxels.cpu().numpy(), coords.cpu().numpy(), num_points.cpu().numpy()
        data_dict['voxels'] = voxels
        data_dict['voxel_coords'] = coords
        data_dict['voxel_num_points'] = num_points
        return data_dict

    def forward(self, data_dict):
        for cur_processor in self.data_processor_queue:
            data_dict = cur_processor(data_dict=data_dict)
        return data_dict
