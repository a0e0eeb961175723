"""Golden vectors of the CenterHead target assignment from the REFERENCE's own class (center_head.py:111-161,202-260), CPU.

    python tests/golden/gen_target_golden.py      (build container only; needs /root/reference)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg          # noqa: E402


def gt_batch():
    """(2, 14, 8): boxes near the range border, a zero-size box, padding rows (class 0 -> 'bg'), all three classes."""
    from detzero_amd.synth import synth_boxes
    a = synth_boxes(91, 12, xy_range=74.0, near_duplicates=0.2)
    b = synth_boxes(92, 9, xy_range=30.0, near_duplicates=0.0)
    out = np.zeros((2, 14, 8), np.float32)
    out[0, :12, :7] = a
    out[0, :12, 7] = np.tile([1, 2, 3], 4)
    out[0, 3, 0:2] = [75.1, -75.19]            # clamped into the last cell
    out[0, 5, 3] = 0.0                          # dx = 0: skipped
    out[1, :9, :7] = b
    out[1, :9, 7] = [1, 1, 2, 3, 3, 2, 1, 2, 3]
    return out


def main():
    from detzero_amd.config import centerpoint_1sweep_cfg
    mods = gg.install_stubs()
    cfg = centerpoint_1sweep_cfg()
    rng = np.array(cfg.DATA_CONFIG.POINT_CLOUD_RANGE, np.float32)
    head = mods['center_head'].CenterHead(cfg.MODEL.DENSE_HEAD, 512, 3, cfg.CLASS_NAMES, np.array([1504, 1504, 40]), rng, [0.1, 0.1, 0.15])
    gt = torch.from_numpy(gt_batch())
    ret = head.assign_targets(gt.clone(), feature_map_size=torch.Size([188, 188]))
    out = {'gt_boxes': gt.numpy(), 'heatmap': ret['heatmaps'][0].numpy(), 'target_boxes': ret['target_boxes'][0].numpy(),
           'inds': ret['inds'][0].numpy(), 'masks': ret['masks'][0].numpy()}
    hm = out['heatmap']
    nz = np.nonzero(hm)
    out['heatmap_nz_idx'] = np.stack(nz, 1).astype(np.int32)
    out['heatmap_nz_val'] = hm[nz]
    del out['heatmap']
    np.savez_compressed(os.path.join(HERE, 'target_golden.npz'), **out)
    print('masks', out['masks'].sum(1), 'nonzero heat-map cells', len(nz[0]))


if __name__ == '__main__':
    main()
