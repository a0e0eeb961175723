#!/usr/bin/env python
"""Golden vectors for a TWO-head CenterHead (center_head.py:81-102 heads_list, :315-385 generate_predicted_boxes): the reference's own
class with CLASS_NAMES_EACH_HEAD = [['Vehicle'], ['Pedestrian', 'Cyclist']] on a random BEV map (CPU, the reference's NMS through
oracle/_ref's iou3d_cpu.cpp as in gen_golden.py).

    python tests/golden/gen_multihead_golden.py        (needs /root/reference; writes tests/golden/multihead_golden.npz)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import gen_golden  # noqa: E402
from detzero_amd.config import AttrDict  # noqa: E402
from detzero_amd.synth import POINT_CLOUD_RANGE, VOXEL_SIZE_02  # noqa: E402
from oracle import voxelize as ov  # noqa: E402


def head_cfg():
    return {
        'CLASS_NAMES_EACH_HEAD': [['Vehicle'], ['Pedestrian', 'Cyclist']], 'SHARED_CONV_CHANNEL': 32,
        'USE_BIAS_BEFORE_NORM': True, 'NUM_HM_CONV': 2, 'IOU_WEIGHT': 1,
        'SEPARATE_HEAD_CFG': {'HEAD_ORDER': ['center', 'center_z', 'dim', 'rot', 'iou'],
                              'HEAD_DICT': {'center': {'out_channels': 2, 'num_conv': 2}, 'center_z': {'out_channels': 1, 'num_conv': 2},
                                            'dim': {'out_channels': 3, 'num_conv': 2}, 'rot': {'out_channels': 2, 'num_conv': 2},
                                            'iou': {'out_channels': 1, 'num_conv': 2}}},
        'TARGET_ASSIGNER_CONFIG': {'FEATURE_MAP_STRIDE': 8},
        'POST_PROCESSING': {'SCORE_THRESH': 0.03, 'POST_CENTER_LIMIT_RANGE': [-80, -80, -10.0, 80, 80, 10.0], 'MAX_OBJ_PER_SAMPLE': 100,
                            'NMS_CONFIG': {'NMS_TYPE': 'nms_gpu', 'NMS_THRESH': 0.7, 'NMS_PRE_MAXSIZE': 4096, 'NMS_POST_MAXSIZE': 500}},
    }


def main():
    mods = gen_golden.install_stubs()
    gen = torch.Generator().manual_seed(23)
    names = ['Vehicle', 'Pedestrian', 'Cyclist']
    grid = ov.grid_size_of(POINT_CLOUD_RANGE, VOXEL_SIZE_02)
    head = mods['center_head'].CenterHead(AttrDict(head_cfg()), 64, 3, names, grid, POINT_CLOUD_RANGE, VOXEL_SIZE_02).eval()
    gen_golden.randomize_bn(head, gen)
    with torch.no_grad():
        for hl in head.heads_list:
            hl.hm[1].bias.fill_(-0.5)
            hl.dim[1].bias.copy_(torch.tensor([1.2, 0.6, 0.4]))
            hl.iou[1].bias.fill_(0.6)
    x = torch.randn((2, 64, 24, 24), generator=gen) * 0.5
    with torch.no_grad():
        s = head.shared_conv(x)
        pds = [hl(s) for hl in head.heads_list]
        boxes = head.generate_predicted_boxes(2, pds)
    out = {'head_' + k: v for k, v in gen_golden.sd_np(head, 'dense_head.').items()}
    out['head_in'] = x.numpy()
    for i, pd in enumerate(pds):
        for k, v in pd.items():
            out['pred%d_%s' % (i, k)] = v.numpy()
    for i, bdict in enumerate(boxes):
        out['boxes_%d' % i] = bdict['pred_boxes'].numpy()
        out['scores_%d' % i] = bdict['pred_scores'].numpy()
        out['labels_%d' % i] = bdict['pred_labels'].numpy()
        print('frame', i, 'final boxes', bdict['pred_boxes'].shape[0], 'labels', np.bincount(bdict['pred_labels'].numpy(), minlength=4)[1:])
    np.savez_compressed(os.path.join(HERE, 'multihead_golden.npz'), **out)


if __name__ == '__main__':
    main()
