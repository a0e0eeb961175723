"""The two static DatasetTemplate hooks the inference loop calls around the model
(/root/reference/detection/detzero_det/datasets/dataset.py:259-354), unchanged in behaviour:
``collate_batch`` (batch index column, padding of gt_boxes) and ``generate_prediction_dicts``
(device -> host, the per-frame record format result.pkl and the tracker consume)."""
from collections import defaultdict

import numpy as np
import torch


def collate_batch(batch_list, _unused=False):
    """dataset.py:259-303, TTA branch included: a sample that is a dict of copies ({'tta_original': ..., 'tta_flip_x': ...})
    contributes the point / voxel arrays of every copy and the remaining keys of the original; ``tta_ops`` lists the copies
    and ``batch_size`` counts frames x copies.  Values may be numpy arrays or device tensors (a frame voxelized on the GPU
    stays there)."""
    data_dict = defaultdict(list)
    tta = 'tta_original' in batch_list[0]
    tta_ops = []
    for cur_sample in batch_list:
        if tta:
            tta_ops = list(cur_sample.keys())
            data_dict['tta_ops'] = list(tta_ops)
            for key in cur_sample['tta_original']:
                if key in ['points', 'voxels', 'voxel_num_points', 'voxel_coords']:
                    for tta_cfg in tta_ops:
                        data_dict[key].append(cur_sample[tta_cfg][key])
                else:
                    data_dict[key].append(cur_sample['tta_original'][key])
        else:
            for key, val in cur_sample.items():
                data_dict[key].append(val)
    batch_size = len(batch_list)
    ret = {}

    def cat(vals):
        return torch.cat(vals, dim=0) if torch.is_tensor(vals[0]) else np.concatenate(vals, axis=0)

    for key, val in data_dict.items():
        if key in ['voxels', 'voxel_num_points']:
            ret[key] = cat(val)
        elif key in ['points', 'voxel_coords']:
            coors = []
            for i, coor in enumerate(val):
                if torch.is_tensor(coor):
                    coors.append(torch.cat([coor.new_full((coor.shape[0], 1), i), coor], dim=1))
                else:
                    coors.append(np.pad(coor, ((0, 0), (1, 0)), mode='constant', constant_values=i))
            ret[key] = cat(coors)
        elif key in ['gt_boxes']:
            max_gt = max(len(x) for x in val)
            out = np.zeros((batch_size, max_gt, val[0].shape[-1]), dtype=np.float32)
            for k in range(batch_size):
                out[k, :len(val[k]), :] = val[k]
            ret[key] = out
        elif key == 'tta_ops':
            ret[key] = val
        else:
            try:
                ret[key] = np.stack(val, axis=0)
            except Exception:
                ret[key] = val
    ret['batch_size'] = batch_size if not tta else int(batch_size * len(tta_ops))
    return ret


def generate_prediction_dicts(batch_dict, pred_dicts, class_names, output_path=None):
    """dataset.py:305-354."""
    annos = []
    for index, box_dict in enumerate(pred_dicts):
        scores = box_dict['pred_scores'].cpu().numpy()
        boxes = box_dict['pred_boxes'].cpu().numpy()
        labels = box_dict['pred_labels'].cpu().numpy()
        n = scores.shape[0]
        rec = {'name': np.zeros(n), 'score': np.zeros(n), 'boxes_lidar': np.zeros([n, 9])}
        if n:
            rec['name'] = np.array(class_names)[labels - 1]
            rec['score'] = scores
            rec['boxes_lidar'] = boxes
        for k in ('sequence_name', 'frame_id', 'pose'):
            if k in batch_dict:
                rec[k] = batch_dict[k][index]
        annos.append(rec)
    return annos
