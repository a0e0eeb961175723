#!/usr/bin/env python
"""Golden vectors for the `roi_features` of the first stage (center_head.py:408-432, 461-486): the reference's own
`CenterHead.get_box_center` (num_point = 5, with the real `box_utils.boxes_to_corners_3d` / `common_utils.rotate_points_along_z`),
`absl_to_relative`, `centernet_utils.bilinear_interpolate_torch` and `reorder_rois_for_refining_features`, run on the CPU over a random
BEV map and boxes that include ones hanging over the map border.

    python tests/golden/gen_roi_feat_golden.py         (needs /root/reference; writes tests/golden/roi_feat_golden.npz)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import gen_golden  # noqa: E402  (the stubs that let the reference's detection modules import on a CPU-only machine)

REF = '/root/reference'


def main():
    mods = gen_golden.install_stubs()
    # the real geometry helpers of the reference (box_utils needs roiaware_pool3d only for other functions)
    cu = gen_golden._load('ref_common_utils', REF + '/utils/detzero_utils/common_utils.py')
    sys.modules['detzero_utils.common_utils'] = cu
    sys.modules['detzero_utils'].common_utils = cu
    sys.modules['detzero_utils.ops.roiaware_pool3d'] = types.ModuleType('detzero_utils.ops.roiaware_pool3d')
    sys.modules['detzero_utils.ops.roiaware_pool3d'].roiaware_pool3d_utils = None
    bu = gen_golden._load('ref_box_utils', REF + '/utils/detzero_utils/box_utils.py')
    ch = mods['center_head']
    ch.boxes_to_corners_3d = bu.boxes_to_corners_3d
    cnu = sys.modules['detzero_det.utils.centernet_utils']

    class Host:                                     # the attributes get_box_center / absl_to_relative read
        num_point = 5
        point_cloud_range = [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0]
        voxel_size = [0.4, 0.4, 0.15]               # 47 x 47 map cells: a small fixture
        feature_map_stride = 8
    host = Host()
    gen = torch.Generator().manual_seed(11)
    b, c, h, w = 2, 16, 47, 47
    bev = torch.randn((b, c, h, w), generator=gen)
    pred_dicts = []
    for n in (37, 0):
        boxes = torch.zeros((n, 7))
        if n:
            boxes[:, 0:2] = (torch.rand((n, 2), generator=gen) - 0.5) * 150.0
            boxes[:5, 0] = torch.tensor([75.1, -75.15, 74.0, -74.9, 80.0])          # on / over the border: clamped corner indices
            boxes[:5, 1] = torch.tensor([-75.1, 75.19, 75.0, 10.0, -80.0])
            boxes[:, 2] = torch.randn(n, generator=gen)
            boxes[:, 3:6] = torch.rand((n, 3), generator=gen) * 6.0 + 0.5
            boxes[:, 6] = (torch.rand(n, generator=gen) - 0.5) * 6.4
        pred_dicts.append({'pred_boxes': boxes, 'pred_scores': torch.rand(n, generator=gen), 'pred_labels': torch.randint(1, 4, (n,), generator=gen)})
    # center_head.py:461-486, verbatim control flow
    features = []
    bev_features = bev.permute(0, 2, 3, 1).contiguous()
    centers_frame = ch.CenterHead.get_box_center(host, pred_dicts)
    ret_maps = []
    for batch_idx in range(b):
        xs, ys = ch.CenterHead.absl_to_relative(host, centers_frame[batch_idx])
        feature_map = cnu.bilinear_interpolate_torch(bev_features[batch_idx], xs, ys)
        if host.num_point > 1:
            section_size = len(feature_map) // host.num_point
            feature_map = torch.cat([feature_map[i * section_size: (i + 1) * section_size] for i in range(host.num_point)], dim=1)
        ret_maps.append(feature_map)
    features.append(ret_maps)
    rois, roi_scores, roi_labels, roi_features = ch.CenterHead.reorder_rois_for_refining_features(b, pred_dicts, features)
    np.savez_compressed(os.path.join(HERE, 'roi_feat_golden.npz'), bev=bev.numpy(), voxel_size=np.array(host.voxel_size), stride=host.feature_map_stride,
                        point_cloud_range=np.array(host.point_cloud_range), boxes0=pred_dicts[0]['pred_boxes'].numpy(),
                        scores0=pred_dicts[0]['pred_scores'].numpy(), labels0=pred_dicts[0]['pred_labels'].numpy(), rois=rois.numpy(),
                        roi_scores=roi_scores.numpy(), roi_labels=roi_labels.numpy(), roi_features=roi_features.numpy())
    print('roi_features', tuple(roi_features.shape), 'max |value| %.3f' % float(roi_features.abs().max()))


if __name__ == '__main__':
    main()
