"""Inference-time mirror of the detector's Waymo dataset (the data formats either side of
the hot path), with the frame assembly on the GPU.

Reference: ``WaymoDetectionDataset`` (detection/detzero_det/datasets/waymo/waymo_dataset.py:15-102) on top of
``DatasetTemplate`` (detection/detzero_det/datasets/dataset.py:22-258): ``ImageSets/<split>.txt`` -> per-sequence info
pickles under ``<root>/<PROCESSED_DATA_TAG>/<sequence>/<sequence>.pkl`` (lists of per-frame dicts with ``lidar_path``,
``sample_idx``, ``sequence_name``, ``sequence_len``, ``pose``, ``time_stamp``[, ``annos``]) -> ``.npy`` sweeps of
(N,6) float32 ``[x,y,z,intensity,elongation,NLZ]`` -> ``merge_sweeps`` -> point feature selection -> data processors.

Here ``__getitem__`` returns the same dict keys (``points``, ``frame_id``, ``pose``, ``sequence_name``,
``use_lead_xyz``), with ``points`` a DEVICE tensor produced by ``dz_merge_sweeps`` (NLZ filter with compaction, tanh,
float64 pose product, time offset) and the column selection of ``absolute_coordinates_encoding``; voxelization stays with
the model side (``FramePipeline``), i.e. the dataset config's ``transform_points_to_voxels_placeholder`` route.
``generate_prediction_dicts`` / ``collate_batch`` are the reference's (dataset_utils.py); ``save_results`` writes the
``result.pkl`` the tracker reads.  Training-time paths (augmentor, gt sampling) and the TF evaluator are not provided.
"""
import copy
import os
import pickle

import numpy as np
import torch

from . import dataset_utils
from . import lib as L
from .lib import DetZeroHipError


def get_sweep_idxs(current_info, sweep_count=(0, 0), current_idx=0):
    """dataset.py:141-162."""
    if not (isinstance(sweep_count, (list, tuple)) and len(sweep_count) == 2):
        raise DetZeroHipError('SWEEP_COUNT must be [lower, upper]')
    cur, n = current_info['sample_idx'], current_info['sequence_len']
    want = [min(max(cur + d, 0), n - 1) for d in range(sweep_count[0], sweep_count[1] + 1)]
    return current_idx + (np.asarray(want) - cur)


def merge_sweeps_gpu(info, target_infos, points, device=None):
    """dataset.py:164-195 on the device.  points: list of (N_i,6) float32 arrays (host) or device tensors.
    Returns ((N',6) float32 device tensor, rows valid up to the returned host count)."""
    dev = device if device is not None else torch.device('cuda', torch.cuda.current_device())
    if len(points) != len(target_infos) or not points:
        raise DetZeroHipError('merge_sweeps: %d sweeps, %d infos' % (len(points), len(target_infos)))
    parts = [p if torch.is_tensor(p) else torch.from_numpy(np.ascontiguousarray(p, dtype=np.float32)) for p in points]
    for p in parts:
        if p.dim() != 2 or p.shape[1] != 6:
            raise DetZeroHipError('merge_sweeps: sweeps are (N,6) [x,y,z,intensity,elongation,NLZ] rows, got %s' % (tuple(p.shape),))
    raw = torch.cat([p.to(dev, torch.float32) for p in parts], dim=0).contiguous()
    offsets = np.zeros(len(parts) + 1, dtype=np.int32)
    np.cumsum([p.shape[0] for p in parts], out=offsets[1:])
    inv_cur = np.linalg.inv(np.asarray(info['pose'], dtype=np.float64))
    mats = np.stack([(inv_cur @ np.asarray(t['pose'], dtype=np.float64))[:3, :].reshape(12) for t in target_infos]).astype(np.float64)
    dts = np.asarray([float(int(t['time_stamp']) - int(info['time_stamp'])) / 1000000. for t in target_infos], dtype=np.float64)
    n = int(offsets[-1])
    lib = L.load()
    out = torch.empty((max(n, 1), 6), dtype=torch.float32, device=dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = torch.empty((lib.dz_merge_sweeps_workspace_bytes(n) // 4 + 1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.dz_merge_sweeps(L.ptr(raw), n, offsets.ctypes.data, mats.ctypes.data, dts.ctypes.data, len(parts), L.ptr(out), L.ptr(cnt),
                                 L.ptr(ws), ws.numel() * 4, L.stream())
    L.check(rc, 'dz_merge_sweeps')
    k = int(cnt.item())
    return out[:k], k


class PointFeatureEncoder:
    """processor/point_feature_encoder.py:6-60, absolute_coordinates_encoding only (the detector configs' choice)."""

    def __init__(self, config, point_cloud_range=None):
        if config.encoding_type != 'absolute_coordinates_encoding':
            raise DetZeroHipError('PointFeatureEncoder: only absolute_coordinates_encoding is provided')
        self.used_feature_list = list(config.used_feature_list)
        self.src_feature_list = list(config.src_feature_list)
        self.point_cloud_range = point_cloud_range
        self.columns = [self.src_feature_list.index(x) for x in self.used_feature_list]

    @property
    def num_point_features(self):
        return len(self.used_feature_list)

    def forward(self, data_dict):
        pts = data_dict['points']
        data_dict['points'] = pts[:, self.columns].contiguous() if torch.is_tensor(pts) else np.concatenate([pts[:, i:i + 1] for i in self.columns], axis=1)
        data_dict['use_lead_xyz'] = True
        return data_dict


class WaymoDetectionDataset(torch.utils.data.Dataset):
    """waymo_dataset.py:15-102 + dataset.py:22-140 for inference (``training=False``)."""

    def __init__(self, dataset_cfg, class_names, root_path=None, training=False, logger=None, device=None):
        if training:
            raise DetZeroHipError('WaymoDetectionDataset: the training path (augmentor, gt sampling) is not provided')
        self.dataset_cfg = dataset_cfg
        self.class_names = list(class_names)
        self.training = False
        self.root_path = str(root_path if root_path is not None else dataset_cfg.DATA_PATH)
        self.logger = logger
        self.device = device
        self.sweep_count = dataset_cfg.get('SWEEP_COUNT', None) or [0, 0]
        self.point_cloud_range = np.array(dataset_cfg.POINT_CLOUD_RANGE, dtype=np.float32)
        self.point_feature_encoder = PointFeatureEncoder(dataset_cfg.POINT_FEATURE_ENCODING, point_cloud_range=self.point_cloud_range)
        # dataset.py:41-50: the config-named processor queue (range mask, voxelization on the device or its placeholder);
        # grid_size / voxel_size are what build_network reads from the dataset
        self.tta = bool(dataset_cfg.get('TTA', False))
        self.test_time_augmentor = None
        if self.tta:                                      # waymo_dataset.py:48-55 (init_tta)
            from .tta import TestTimeAugmentor
            self.test_time_augmentor = TestTimeAugmentor(dataset_cfg.TEST_TIME_AUGMENTOR, logger=logger)
        self.data_processor = None
        self.grid_size = self.voxel_size = None
        if dataset_cfg.get('DATA_PROCESSOR', None):
            from .data_processor import DataProcessor
            self.data_processor = DataProcessor(dataset_cfg.DATA_PROCESSOR, point_cloud_range=self.point_cloud_range, training=False,
                                                num_point_features=self.point_feature_encoder.num_point_features)
            self.grid_size, self.voxel_size = self.data_processor.grid_size, self.data_processor.voxel_size
        self.data_path = self.root_path + '/' + dataset_cfg.PROCESSED_DATA_TAG
        self.split = dataset_cfg.DATA_SPLIT[self.mode]
        self.infos = []
        self._read_split()
        self.init_infos()

    @property
    def mode(self):
        return 'test'

    def _log(self, msg):
        if self.logger is not None:
            self.logger.info(msg)

    def _read_split(self):
        with open(os.path.join(self.root_path, 'ImageSets', self.split + '.txt')) as f:
            self.sample_sequence_list = [x.strip() for x in f.readlines()]

    def set_split(self, split):
        self.split = split
        self._read_split()
        self.infos = []
        self.init_infos()

    @staticmethod
    def check_sequence_name_with_all_version(seq_file):
        """waymo_dataset.py:85-91, verbatim in behaviour: written for `<name>.tfrecord` paths (the 9 characters it strips);
        on the `.pkl` paths init_infos passes it only the second rule (drop the suffix) does something useful."""
        if '_with_camera_labels' not in seq_file and not os.path.exists(seq_file):
            seq_file = seq_file[:-9] + '_with_camera_labels.tfrecord'
        if '_with_camera_labels' in seq_file and not os.path.exists(seq_file):
            seq_file = seq_file.replace('_with_camera_labels', '')
        return seq_file

    def init_infos(self):
        """waymo_dataset.py:57-83."""
        infos, skipped = [], 0
        for name in self.sample_sequence_list:
            sequence_name = os.path.splitext(name)[0]
            info_path = self.check_sequence_name_with_all_version(os.path.join(self.data_path, sequence_name, '%s.pkl' % sequence_name))
            if not os.path.exists(info_path):
                skipped += 1
                continue
            with open(info_path, 'rb') as f:
                infos.extend(pickle.load(f))
        self.infos.extend(infos)
        self._log('Total skipped info %s' % skipped)
        self._log('Total samples for Waymo dataset: %d' % len(infos))
        interval = self.dataset_cfg.SAMPLED_INTERVAL[self.mode] if 'SAMPLED_INTERVAL' in self.dataset_cfg else 1
        if interval > 1:
            self.infos = self.infos[::interval]
            self._log('Total sampled samples for Waymo dataset: %d' % len(self.infos))

    def __len__(self):
        return len(self.infos)

    def get_infos_and_points(self, idx_list):
        """waymo_dataset.py:93-102."""
        infos, points = [], []
        for i in idx_list:
            infos.append(self.infos[i])
            points.append(np.load(self.infos[i]['lidar_path']))
        return infos, points

    def __getitem__(self, index):
        """dataset.py:106-132,197-258 (inference branch; gt boxes are passed through untouched when the infos carry them)."""
        current_info = copy.deepcopy(self.infos[index])
        # NB like the reference, sweeps are looked up by position in self.infos (valid for SAMPLED_INTERVAL 1)
        target_infos, points = self.get_infos_and_points(get_sweep_idxs(current_info, self.sweep_count, index))
        merged, _ = merge_sweeps_gpu(current_info, target_infos, points, self.device)
        data_dict = {'points': merged, 'frame_id': current_info['sample_idx'], 'pose': current_info['pose'],
                     'sequence_name': current_info['sequence_name']}
        if 'annos' in current_info:
            annos = current_info['annos']
            keep = [i for i, n in enumerate(annos['name']) if n != 'unknown']
            names = np.asarray(annos['name'])[keep]
            sel = [i for i, n in enumerate(names) if n in self.class_names]
            boxes = np.asarray(annos['gt_boxes_lidar'])[keep][sel]
            classes = np.array([self.class_names.index(n) + 1 for n in names[sel]], dtype=np.int32)
            data_dict['gt_boxes'] = np.concatenate((boxes, classes.reshape(-1, 1).astype(np.float32)), axis=1)
        data_dict = self.point_feature_encoder.forward(data_dict)
        if self.tta:                                      # dataset.py:239-246: every copy goes through the processors
            copies = self.test_time_augmentor.forward(data_dict)
            if self.data_processor is not None:
                copies = {k: self.data_processor.forward(data_dict=v) for k, v in copies.items()}
            return copies
        if self.data_processor is not None:               # dataset.py:249-251
            data_dict = self.data_processor.forward(data_dict=data_dict)
        return data_dict

    def evaluation(self, det_annos, class_names, **kwargs):
        """waymo_dataset.py:104-131.  Without ground truth in the infos the reference returns early; with it, it calls the
        TensorFlow / waymo_open_dataset metrics, which are outside this backend (recall statistics come from the model's
        generate_recall_record)."""
        if not self.infos or 'annos' not in self.infos[0]:
            return 'No ground-truth boxes for evaluation', {}
        return 'Official Waymo metrics (waymo_open_dataset / TensorFlow) are not provided by the HIP backend', {}

    collate_batch = staticmethod(dataset_utils.collate_batch)
    generate_prediction_dicts = staticmethod(dataset_utils.generate_prediction_dicts)

    @staticmethod
    def save_results(det_annos, output_dir, name='result.pkl'):
        """tools/test.py / eval_utils: the list of per-frame prediction dicts, pickled for the tracker."""
        os.makedirs(output_dir, exist_ok=True)
        path = os.path.join(output_dir, name)
        with open(path, 'wb') as f:
            pickle.dump(det_annos, f)
        return path
