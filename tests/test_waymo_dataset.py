"""Waymo dataset mirror: sweep selection / merge oracle against the reference's own static methods (CPU), the file layout
logic on a synthetic on-disk dataset (CPU), the device merge and the whole __getitem__ -> detector path (GPU)."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch

from detzero_amd.config import AttrDict
from oracle import waymo_io as oracle_io

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import gen_waymo_io_golden as gen          # noqa: E402  (case list + synthetic sequences; importing it does not touch the reference)


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'waymo_io_golden.npz'))


def _dataset_cfg(sweep_count=None, interval=1):
    src = ['x', 'y', 'z', 'intensity', 'elongation', 'offset']
    cfg = {'DATA_PATH': '', 'PROCESSED_DATA_TAG': 'waymo_processed_data', 'DATA_SPLIT': {'train': 'train', 'test': 'val'},
           'SAMPLED_INTERVAL': {'train': 1, 'test': interval}, 'POINT_CLOUD_RANGE': [-75.2, -75.2, -2, 75.2, 75.2, 4],
           'POINT_FEATURE_ENCODING': AttrDict({'encoding_type': 'absolute_coordinates_encoding',
                                               'used_feature_list': src[:5] if sweep_count is None else src, 'src_feature_list': src})}
    if sweep_count is not None:
        cfg['SWEEP_COUNT'] = sweep_count
    return AttrDict(cfg)


def _write_dataset(root, seeds=(1, 2), with_suffix=(False, True)):
    """<root>/ImageSets/val.txt + <root>/waymo_processed_data/<seq>/{<seq>.pkl, 0000.npy ...} as waymo_preprocess.py lays it out."""
    os.makedirs(os.path.join(root, 'ImageSets'))
    names = []
    for seed, suffix in zip(seeds, with_suffix):
        infos, sweeps = gen.synth_sequence(seed, n_frames=4, n_points=500)
        seq = 'segment-%d%s' % (1000 + seed, '_with_camera_labels' if suffix else '')
        d = os.path.join(root, 'waymo_processed_data', seq)
        os.makedirs(d)
        for i, (info, pts) in enumerate(zip(infos, sweeps)):
            info['sequence_name'] = seq
            info['lidar_path'] = os.path.join(d, '%04d.npy' % i)
            np.save(info['lidar_path'], pts)
        with open(os.path.join(d, seq + '.pkl'), 'wb') as f:
            pickle.dump(infos, f)
        names.append(seq + '.tfrecord')                                      # the split file names the tfrecord of the sequence
    names.append('segment-9999.tfrecord')                                   # not on disk: skipped
    with open(os.path.join(root, 'ImageSets', 'val.txt'), 'w') as f:
        f.write('\n'.join(names) + '\n')


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize('seed,sweep_count,idx', gen.CASES)
def test_oracle_sweeps_equal_reference(g, seed, sweep_count, idx):
    from detzero_amd.waymo_dataset import get_sweep_idxs
    infos, sweeps = gen.synth_sequence(seed)
    tl = oracle_io.get_sweep_idxs(infos[idx], sweep_count, idx)
    np.testing.assert_array_equal(tl, g['c%d_idx' % seed])
    np.testing.assert_array_equal(get_sweep_idxs(infos[idx], sweep_count, idx), g['c%d_idx' % seed])
    merged = oracle_io.merge_sweeps(infos[idx], [infos[i] for i in tl], [sweeps[i].copy() for i in tl])
    np.testing.assert_array_equal(merged, g['c%d_points' % seed])           # bit for bit, float64


def test_dataset_file_layout(tmp_path):
    from detzero_amd.waymo_dataset import WaymoDetectionDataset
    root = str(tmp_path / 'waymo')
    _write_dataset(root)
    ds = WaymoDetectionDataset(_dataset_cfg(), ['Vehicle', 'Pedestrian', 'Cyclist'], root_path=root)
    assert len(ds) == 8 and ds.mode == 'test' and ds.point_feature_encoder.num_point_features == 5
    assert [i['sample_idx'] for i in ds.infos] == [0, 1, 2, 3, 0, 1, 2, 3]
    infos, pts = ds.get_infos_and_points([5])
    assert infos[0]['sequence_name'].endswith('_with_camera_labels') and pts[0].shape == (500, 6)
    ds2 = WaymoDetectionDataset(_dataset_cfg(interval=3), ['Vehicle'], root_path=root)
    assert len(ds2) == 3
    path = ds.save_results([{'name': np.array(['Vehicle']), 'score': np.array([0.5]), 'boxes_lidar': np.zeros((1, 7)), 'frame_id': 0}],
                           str(tmp_path / 'out'))
    assert pickle.load(open(path, 'rb'))[0]['frame_id'] == 0


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize('seed,sweep_count,idx', gen.CASES)
def test_merge_sweeps_device_matches_reference(device, g, seed, sweep_count, idx):
    from detzero_amd.waymo_dataset import merge_sweeps_gpu
    infos, sweeps = gen.synth_sequence(seed)
    tl = g['c%d_idx' % seed]
    out, k = merge_sweeps_gpu(infos[idx], [infos[i] for i in tl], [sweeps[i] for i in tl], device)
    ref = g['c%d_points' % seed]
    assert k == ref.shape[0]
    got = out.cpu().numpy()
    np.testing.assert_array_equal(got[:, [0, 1, 2, 4]], ref[:, [0, 1, 2, 4]].astype(np.float32))      # coordinates: one float32 rounding of the float64 product
    np.testing.assert_allclose(got[:, 3], ref[:, 3], rtol=0, atol=2e-7)                               # tanhf vs numpy's float32 tanh
    np.testing.assert_array_equal(got[:, 5], ref[:, 5].astype(np.float32))


@pytest.mark.gpu
def test_dataset_to_detector(device, tmp_path):
    """ImageSets + info pickles + .npy sweeps -> WaymoDetectionDataset.__getitem__ (device points) -> FramePipeline -> result.pkl."""
    from detzero_amd.centerpoint import FramePipeline
    from detzero_amd.synth import VOXEL_SIZE_02
    from detzero_amd.waymo_dataset import WaymoDetectionDataset
    from tests.util import make_model
    root = str(tmp_path / 'waymo')
    _write_dataset(root)
    ds = WaymoDetectionDataset(_dataset_cfg(), ['Vehicle', 'Pedestrian', 'Cyclist'], root_path=root, device=device)
    item = ds[2]
    infos, sweeps = gen.synth_sequence(1, n_frames=4, n_points=500)
    ref = oracle_io.merge_sweeps(infos[2], [infos[2]], [sweeps[2].copy()])[:, :5]
    assert item['points'].is_cuda and item['use_lead_xyz'] and item['frame_id'] == 2
    np.testing.assert_allclose(item['points'].cpu().numpy(), ref, rtol=0, atol=2e-7)
    ds6 = WaymoDetectionDataset(_dataset_cfg(sweep_count=[-1, 0]), ['Vehicle'], root_path=root, device=device)
    assert ds6[1]['points'].shape[1] == 6
    model, cfg, info = make_model(VOXEL_SIZE_02, seed=0, gain='default')      # plumbing test on 500-point frames: the weight set that emits boxes for any input
    pipe = FramePipeline(model.to(device), info)
    batch = ds.collate_batch([ds[0], ds[1]])
    boxes, counts = pipe([ds[0]['points'], ds[1]['points']])
    preds = [{'pred_boxes': boxes[i, :int(counts[i]), :7], 'pred_scores': boxes[i, :int(counts[i]), 7],
              'pred_labels': boxes[i, :int(counts[i]), 8].long()} for i in range(2)]
    annos = ds.generate_prediction_dicts(batch, preds, ds.class_names)
    assert len(annos) == 2 and annos[1]['frame_id'] == 1 and annos[0]['boxes_lidar'].shape[1] == 7
    assert pickle.load(open(ds.save_results(annos, str(tmp_path / 'res')), 'rb'))[0]['sequence_name'] == batch['sequence_name'][0]
