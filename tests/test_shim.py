"""The plugin boundary, shown rather than asserted: the reference's own ``detection/tools/test.py`` /
``eval_utils.py`` import and run against the ``detzero_det`` / ``detzero_utils`` shim packages (detzero_amd/shim).

CPU: the reference files are loaded BY PATH from /root/reference where that tree exists (this container; skipped elsewhere) -
imports resolve, ``parse_config`` reads the reference's own yaml, ``build_dataloader`` / ``build_network`` accept what it
passes, and the reference's ``eval_one_epoch`` and the shim's restated loop write identical ``result.pkl`` files over a stub
model.  GPU: the loop (the reference's file when present, else the restated one) drives the HIP ``CenterPoint`` over an
on-disk synthetic Waymo dataset with ground truth - boxes equal to FramePipeline's, recall record checked against the oracle.
"""
import importlib.util
import logging
import os
import pickle
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from detzero_amd import shim
from detzero_amd.config import AttrDict, centerpoint_1sweep_cfg

REF_TOOLS = '/root/reference/detection/tools'
has_ref = os.path.isfile(os.path.join(REF_TOOLS, 'test.py'))
CLASS_NAMES = ['Vehicle', 'Pedestrian', 'Cyclist']


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _reference_eval_utils():
    shim.install()
    return _load_by_path('eval_utils', os.path.join(REF_TOOLS, 'eval_utils.py'))


def test_shim_packages_export_the_import_surface():
    """test.py:1-19, eval_utils.py:1-11, train.py:1-20: every name they import exists."""
    shim.install()
    from detzero_utils import common_utils, config_utils, model_utils
    from detzero_det import datasets, models
    for fn in ('create_logger', 'init_dist_pytorch', 'init_dist_slurm', 'get_dist_info', 'merge_results_dist', 'set_random_seed'):
        assert callable(getattr(common_utils, fn)), fn
    for fn in ('cfg_from_list', 'cfg_from_yaml_file', 'log_config_to_file'):
        assert callable(getattr(config_utils, fn)), fn
    assert config_utils.cfg.LOCAL_RANK == 0
    assert callable(model_utils.load_params_from_file) and callable(model_utils.load_params_with_optimizer)
    assert callable(datasets.build_dataloader) and 'WaymoDetectionDataset' in datasets.__all__
    assert callable(models.build_network) and callable(models.load_data_to_gpu) and callable(models.model_fn_decorator)
    from tensorboardX import SummaryWriter
    from easydict import EasyDict
    assert EasyDict({'a': {'b': 1}}).a.b == 1 and SummaryWriter is not None


def test_cfg_from_list_and_logging(tmp_path):
    shim.install()
    from detzero_utils import common_utils, config_utils
    cfg = AttrDict({'A': {'B': 1, 'C': [1.0, 2.0], 'D': {'x': 1, 'y': 2.5}, 'N': ['u']}, 'S': 'txt'})
    config_utils.cfg_from_list(['A.B', '7', 'A.C', '[3.0, 4.0]', 'A.D', 'x:5,y:0.5', 'S', 'other', 'A.N', 'p,q'], cfg)
    assert cfg.A.B == 7 and cfg.A.C == [3.0, 4.0] and cfg.A.N == ['p', 'q'] and cfg.A.D.x == 5 and cfg.A.D.y == 0.5 and cfg.S == 'other'
    with pytest.raises(AssertionError):
        config_utils.cfg_from_list(['A.MISSING', '1'], cfg)
    logger = common_utils.create_logger(str(tmp_path / 'log.txt'))
    config_utils.log_config_to_file(cfg, logger=logger)
    for h in list(logger.handlers):
        h.flush()
    assert 'cfg.A.B: 7' in open(tmp_path / 'log.txt').read()
    assert common_utils.get_dist_info() == (0, 1)


def test_checkpoint_loading_and_sparse_layout(tmp_path):
    """load_params_from_file: reference-layout checkpoints load completely; the (kD,kH,kW,Cin,Cout) sparse-conv layout is
    converted instead of being skipped silently; a checkpoint with no usable backbone weight raises."""
    shim.install()
    from detzero_utils import model_utils
    from detzero_amd.centerpoint import synth_detector
    from detzero_amd.lib import DetZeroHipError
    from detzero_amd.synth import VOXEL_SIZE_01
    model, cfg, info = synth_detector(VOXEL_SIZE_01, seed=5)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    logger = logging.getLogger('shim-test')
    torch.save({'model_state': sd, 'version': 'x'}, tmp_path / 'a.pth')
    other, _, _ = synth_detector(VOXEL_SIZE_01, seed=6)
    model_utils.load_params_from_file(other, str(tmp_path / 'a.pth'), logger, to_cpu=True)
    assert all(torch.equal(v, sd[k]) for k, v in other.state_dict().items())
    native = {k: (v.permute(1, 2, 3, 4, 0).contiguous() if v.dim() == 5 else v) for k, v in sd.items()}
    torch.save({'model_state': native}, tmp_path / 'b.pth')
    other2, _, _ = synth_detector(VOXEL_SIZE_01, seed=7)
    model_utils.load_params_from_file(other2, str(tmp_path / 'b.pth'), logger, to_cpu=True)
    assert all(torch.equal(v, sd[k]) for k, v in other2.state_dict().items())
    bad = {k: (v[..., :1] if v.dim() == 5 else v) for k, v in sd.items() if not k.startswith('backbone3d') or v.dim() == 5}
    torch.save({'model_state': bad}, tmp_path / 'c.pth')
    with pytest.raises(DetZeroHipError):
        model_utils.load_params_from_file(other2, str(tmp_path / 'c.pth'), logger, to_cpu=True)
    with pytest.raises(FileNotFoundError):
        model_utils.load_params_from_file(other2, str(tmp_path / 'none.pth'), logger)


# ------------------------------------------------------------------------------------------------ synthetic on-disk dataset
def _dataset_cfg(root, with_processor=True):
    cfg = centerpoint_1sweep_cfg((0.2, 0.2, 0.15))
    src = ['x', 'y', 'z', 'intensity', 'elongation']
    cfg.DATA_CONFIG.update(AttrDict({'DATASET': 'WaymoDetectionDataset', 'DATA_PATH': root, 'PROCESSED_DATA_TAG': 'waymo_processed_data',
                                     'DATA_SPLIT': {'train': 'train', 'test': 'val'}, 'SAMPLED_INTERVAL': {'train': 1, 'test': 1},
                                     'POINT_FEATURE_ENCODING': {'encoding_type': 'absolute_coordinates_encoding',
                                                                'used_feature_list': src, 'src_feature_list': src + ['nlz']}}))
    cfg.LOCAL_RANK = 0
    return cfg


def _write_dataset(root, n_frames=4, n_points=20000, with_annos=True):
    """waymo_preprocess.py layout: ImageSets/val.txt, <tag>/<seq>/<seq>.pkl + %04d.npy (N,6) frames; annos carry gt boxes."""
    from detzero_amd.synth import synth_boxes, synth_waymo_frame
    os.makedirs(os.path.join(root, 'ImageSets'))
    seq = 'segment-777_with_camera_labels'
    d = os.path.join(root, 'waymo_processed_data', seq)
    os.makedirs(d)
    infos = []
    for i in range(n_frames):
        p5 = synth_waymo_frame(300 + i, n_points)
        p5[:, 3] = np.arctanh(np.clip(p5[:, 3], 0, 0.999))                                  # raw intensity: the dataset applies tanh
        pts = np.concatenate([p5, -np.ones((n_points, 1), np.float32)], 1).astype(np.float32)
        path = os.path.join(d, '%04d.npy' % i)
        np.save(path, pts)
        info = {'sample_idx': i, 'sequence_len': n_frames, 'sequence_name': seq, 'pose': np.eye(4), 'time_stamp': 1000000 * i, 'lidar_path': path}
        if with_annos:
            gt = synth_boxes(40 + i, 12, xy_range=60.0, near_duplicates=0.0)
            info['annos'] = {'name': np.array(['Vehicle', 'Pedestrian', 'unknown', 'Cyclist'] * 3), 'gt_boxes_lidar': gt}
        infos.append(info)
    with open(os.path.join(d, seq + '.pkl'), 'wb') as f:
        pickle.dump(infos, f)
    with open(os.path.join(root, 'ImageSets', 'val.txt'), 'w') as f:
        f.write(seq + '.tfrecord\n')
    return infos


# ------------------------------------------------------------------------------------------------ reference tools, CPU
@pytest.mark.skipif(not has_ref, reason='reference tree not present')
def test_reference_test_py_imports_and_parses_config(tmp_path, monkeypatch):
    """detection/tools/test.py, unmodified and loaded by path: its imports resolve against the shims and its parse_config reads
    the reference's own centerpoint_1sweep.yaml (+ the _BASE_CONFIG_ include) into the global cfg; the shim's build_dataloader
    and build_network accept exactly what its main() passes (test.py:186-196)."""
    shim.install()
    monkeypatch.chdir(REF_TOOLS)
    monkeypatch.syspath_prepend(REF_TOOLS)
    monkeypatch.setattr(sys, 'argv', ['test.py', '--cfg_file', 'cfgs/det_model_cfgs/centerpoint_1sweep.yaml', '--batch_size', '2',
                                      '--set', 'MODEL.DENSE_HEAD.POST_PROCESSING.SCORE_THRESH', '0.05'])
    for m in ('test', 'eval_utils'):
        sys.modules.pop(m, None)
    ref_test = _load_by_path('ref_tools_test', os.path.join(REF_TOOLS, 'test.py'))
    from detzero_utils.config_utils import cfg as gcfg
    for k in list(gcfg.keys()):
        if k != 'LOCAL_RANK':
            del gcfg[k]
    args, cfg = ref_test.parse_config()
    assert cfg.MODEL.NAME == 'CenterPoint' and cfg.DATA_CONFIG.DATASET == 'WaymoDetectionDataset' and cfg.TAG == 'centerpoint_1sweep'
    assert cfg.MODEL.DENSE_HEAD.POST_PROCESSING.SCORE_THRESH == 0.05 and cfg.OPTIMIZATION.BATCH_SIZE_PER_GPU == 8
    assert cfg.DATA_CONFIG.DATA_PROCESSOR[2].NAME == 'transform_points_to_voxels'
    root = str(tmp_path / 'waymo')
    _write_dataset(root, n_frames=3, n_points=2000)
    logger = ref_test.common_utils.create_logger(None, rank=0)
    test_set, test_loader, sampler = ref_test.build_dataloader(dataset_cfg=cfg.DATA_CONFIG, class_names=cfg.CLASS_NAMES, batch_size=args.batch_size,
                                                               dist=False, workers=args.workers, logger=logger, training=False, root_path=root)
    assert len(test_set) == 3 and len(test_loader) == 2 and sampler is None and tuple(test_set.grid_size) == (1504, 1504, 40)
    model = ref_test.build_network(model_cfg=cfg.MODEL, num_class=len(cfg.CLASS_NAMES), dataset=test_set)
    assert type(model).__name__ == 'CenterPoint' and model.backbone3d.sparse_shape == [41, 1504, 1504]
    assert callable(ref_test.eval_utils.eval_one_epoch)


class _StubDataset(torch.utils.data.Dataset):
    """What eval_one_epoch touches of a dataset (eval_utils.py:56-58,94-97,137-141), without a device."""
    from detzero_amd.dataset_utils import collate_batch as _cb, generate_prediction_dicts as _gp
    collate_batch = staticmethod(_cb)
    generate_prediction_dicts = staticmethod(_gp)
    class_names = CLASS_NAMES

    def __len__(self):
        return 5

    def __getitem__(self, i):
        return {'frame_id': i, 'sequence_name': 'seq', 'pose': np.eye(4) * (i + 1)}

    def evaluation(self, det_annos, class_names, **kwargs):
        return 'stub evaluation of %d frames' % len(det_annos), {'n': len(det_annos)}


class _StubModel(torch.nn.Module):
    def forward(self, batch_dict):
        preds = []
        for fid in batch_dict['frame_id']:
            n = int(fid) % 3 + 1
            g = torch.Generator().manual_seed(int(fid))
            preds.append({'pred_boxes': torch.rand((n, 7), generator=g), 'pred_scores': torch.rand((n,), generator=g),
                          'pred_labels': torch.randint(1, 4, (n,), generator=g)})
        return preds, {'gt': 2 * len(preds), 'rcnn_0.3': len(preds), 'rcnn_0.5': 1, 'rcnn_0.7': 0}


def _stub_cfg():
    cfg = centerpoint_1sweep_cfg()
    cfg.LOCAL_RANK = 0
    return cfg


@pytest.mark.skipif(not has_ref, reason='reference tree not present')
def test_reference_eval_loop_equals_restated_loop(tmp_path):
    """The reference's eval_utils.eval_one_epoch (by path) and tests/eval_loop.py over the same loader and model:
    identical result.pkl, identical recall dictionary."""
    ref_eval = _reference_eval_utils()
    from tests import eval_loop as my_eval
    ds = _StubDataset()
    loader = torch.utils.data.DataLoader(ds, batch_size=2, collate_fn=ds.collate_batch)
    logger = logging.getLogger('shim-eval')
    r1 = ref_eval.eval_one_epoch(_stub_cfg(), _StubModel(), loader, 'x', logger, result_dir=Path(tmp_path / 'ref'), save_tb=True)
    r2 = my_eval.eval_one_epoch(_stub_cfg(), _StubModel(), loader, 'x', logger, result_dir=Path(tmp_path / 'mine'))
    assert r1 == r2 and r1['n'] == 5 and r1['recall/rcnn_0.3'] == 0.5
    a = pickle.load(open(tmp_path / 'ref' / 'result.pkl', 'rb'))
    b = pickle.load(open(tmp_path / 'mine' / 'result.pkl', 'rb'))
    assert len(a) == len(b) == 5
    for x, y in zip(a, b):
        assert sorted(x) == sorted(y)
        for k in x:
            assert np.array_equal(np.asarray(x[k]), np.asarray(y[k])), k


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_eval_one_epoch_drives_the_hip_detector(device, tmp_path):
    """build_dataloader -> build_network -> eval_one_epoch (the reference's file where the tree exists, else the restated loop)
    over an on-disk dataset WITH ground truth: result.pkl equals FramePipeline's boxes, the recall record (C5,
    centerpoint.py:310-352) equals an independent count from the oracle's BEV overlap."""
    shim.install()
    from detzero_det.datasets import build_dataloader
    from detzero_det.models import build_network
    from detzero_amd.centerpoint import FramePipeline, SyntheticDatasetInfo, synth_detector
    from detzero_amd.synth import VOXEL_SIZE_02
    from oracle import cref
    ev = _reference_eval_utils() if has_ref else __import__('tests.eval_loop', fromlist=['x'])
    root = str(tmp_path / 'waymo')
    infos = _write_dataset(root)
    cfg = _dataset_cfg(root)
    logger = logging.getLogger('shim-gpu')
    ds, loader, _ = build_dataloader(dataset_cfg=cfg.DATA_CONFIG, class_names=cfg.CLASS_NAMES, batch_size=2, dist=False, workers=4,
                                     logger=logger, training=False)
    assert tuple(ds.grid_size) == (752, 752, 40) and len(loader) == 2
    ref_model, _, info = synth_detector(VOXEL_SIZE_02, seed=0)
    model = build_network(model_cfg=cfg.MODEL, num_class=3, dataset=ds)
    model.load_state_dict(ref_model.state_dict())
    model.cuda()
    ret = ev.eval_one_epoch(cfg, model, loader, 'e0', logger, result_dir=Path(tmp_path / 'out'), save_to_file=False)
    annos = pickle.load(open(tmp_path / 'out' / 'result.pkl', 'rb'))
    assert len(annos) == 4 and [a['frame_id'] for a in annos] == [0, 1, 2, 3]
    pipe = FramePipeline(ref_model.to(device), SyntheticDatasetInfo(cfg))
    hits = {0.3: 0, 0.5: 0, 0.7: 0}
    n_gt = 0
    for i, a in enumerate(annos):
        item = ds[i]
        assert item['voxels'].is_cuda and item['voxel_coords'].shape[1] == 3
        out, d_n = pipe(item['points'])
        n = int(d_n.item())
        assert a['boxes_lidar'].shape == (n, 7) and n > 0
        np.testing.assert_array_equal(a['boxes_lidar'], out[:n, :7].cpu().numpy())
        np.testing.assert_array_equal(a['score'], out[:n, 7].cpu().numpy())
        assert list(a['name']) == [CLASS_NAMES[int(l) - 1] for l in out[:n, 8].cpu().numpy()]
        assert a['sequence_name'] == infos[i]['sequence_name'] and np.array_equal(a['pose'], np.eye(4))
        # independent recall count: 3-D IoU from the oracle's rotated BEV overlap (pinned on the reference's iou3d_cpu.cpp)
        gt = item['gt_boxes'][:, :7].astype(np.float32)
        assert gt.shape[0] == 9 and set(item['gt_boxes'][:, 7].astype(int)) == {1, 2, 3}        # 'unknown' dropped, classes 1-based
        pb = a['boxes_lidar'].astype(np.float32)
        ov = cref.boxes_overlap_bev(pb, gt)
        top = np.minimum(pb[:, None, 2] + pb[:, None, 5] / 2, gt[None, :, 2] + gt[None, :, 5] / 2)
        bot = np.maximum(pb[:, None, 2] - pb[:, None, 5] / 2, gt[None, :, 2] - gt[None, :, 5] / 2)
        o3 = ov * np.clip(top - bot, 0, None)
        iou = o3 / np.clip((pb[:, 3] * pb[:, 4] * pb[:, 5])[:, None] + (gt[:, 3] * gt[:, 4] * gt[:, 5])[None, :] - o3, 1e-6, None)
        best = iou.max(0)
        assert not np.any(np.abs(best[:, None] - np.array([0.3, 0.5, 0.7])[None, :]) < 1e-4)    # no gt sits on a threshold
        for t in hits:
            hits[t] += int((best > t).sum())
        n_gt += gt.shape[0]
    for t in hits:
        assert ret['recall/rcnn_%s' % t] == pytest.approx(hits[t] / n_gt), (t, ret, hits)
    assert n_gt == 36


@pytest.mark.gpu
def test_generate_recall_record_matches_reference_golden(device, golden_dir):
    """C5 against the reference's OWN generate_recall_record (centerpoint.py:310-352) and boxes_iou3d_gpu, executed on CPU by
    tests/golden/gen_recall_golden.py (only the CUDA overlap call is replaced, by the routine pinned on iou3d_cpu.cpp): the
    accumulated record after each of four frames - padded ground truth, a frame without detections, a second-stage frame with rois."""
    from detzero_amd.centerpoint import CenterPoint
    g = np.load(os.path.join(golden_dir, 'recall_golden.npz'))
    thresh = [float('%g' % t) for t in g['thresh']]
    rec = {}
    for ci in range(int(g['n_cases'])):
        data = {'gt_boxes': torch.from_numpy(g['gt%d' % ci]).to(device)[None]}
        if 'rois%d' % ci in g.files:
            data['rois'] = torch.from_numpy(g['rois%d' % ci]).to(device)[None]
        rec = CenterPoint.generate_recall_record(torch.from_numpy(g['pred%d' % ci]).to(device), rec, 0, data, thresh)
        got = [rec['gt']] + [rec['roi_%s' % t] for t in thresh] + [rec['rcnn_%s' % t] for t in thresh]
        assert got == g['recall_after%d' % ci].tolist(), (ci, rec)
    assert rec['roi_0.3'] > 0 and rec['rcnn_0.7'] > 0
    assert CenterPoint.generate_recall_record(torch.from_numpy(g['pred0']).to(device), {}, 0, {}, [0.3]) == {}


@pytest.mark.gpu
def test_tta_through_the_plugin_surface(device, tmp_path):
    """TTA: True in the dataset config (waymo_1sweep.yaml:31,48-58): the dataset emits the dict of augmented copies, collate_batch
    stacks them as frames x copies, CenterPoint.post_processing restores and fuses them (centerpoint.py:131-208,298-306) - and
    the result equals TTAPipeline over the same frame."""
    shim.install()
    from detzero_det.datasets import build_dataloader
    from detzero_det.models import build_network
    from detzero_amd import tta
    from detzero_amd.centerpoint import FramePipeline, SyntheticDatasetInfo, synth_detector
    from detzero_amd.synth import VOXEL_SIZE_02
    root = str(tmp_path / 'waymo')
    _write_dataset(root, n_frames=2, with_annos=False)
    cfg = _dataset_cfg(root)
    cfg.DATA_CONFIG.TTA = True
    cfg.DATA_CONFIG.TEST_TIME_AUGMENTOR = AttrDict({'DISABLE_AUG_LIST': ['placeholder'], 'AUG_CONFIG_LIST': [
        {'NAME': 'world_flip', 'ALONG_AXIS_LIST': ['x', 'y']}, {'NAME': 'world_rotation', 'ROT_ANGLE': [0, 0.39365818, -0.78539816]},
        {'NAME': 'world_scaling', 'SCALE_RANGE': [0.95, 1.05]}]})
    logger = logging.getLogger('shim-tta')
    ds, loader, _ = build_dataloader(dataset_cfg=cfg.DATA_CONFIG, class_names=cfg.CLASS_NAMES, batch_size=1, dist=False, workers=0,
                                     logger=logger, training=False)
    assert ds.tta and ds.test_time_augmentor.op_names == ['tta_original', 'tta_flip_x', 'tta_flip_y', 'tta_rot_0.39365818', 'tta_rot_-0.78539816',
                                                          'tta_scale_0.95', 'tta_scale_1.05']
    ref_model, _, info = synth_detector(VOXEL_SIZE_02, seed=0)
    model = build_network(model_cfg=cfg.MODEL, num_class=3, dataset=ds)
    model.load_state_dict(ref_model.state_dict())
    model.cuda().eval()
    batch = next(iter(loader))
    assert batch['batch_size'] == 7 and batch['tta_ops'][2] == 'tta_flip_y' and int(batch['voxel_coords'][:, 0].max()) == 6
    from detzero_det.models import load_data_to_gpu
    load_data_to_gpu(batch)
    with torch.no_grad():
        pred_dicts, _ = model(batch)
    assert len(pred_dicts) == 1 and batch['batch_size'] == 1
    got_b, got_s, got_l = pred_dicts[0]['pred_boxes'], pred_dicts[0]['pred_scores'], pred_dicts[0]['pred_labels']
    assert got_b.shape[0] > 10 and got_b.dtype == torch.float64 and bool((got_s[:-1] >= got_s[1:]).all())
    pipe = FramePipeline(ref_model.to(device), SyntheticDatasetInfo(cfg))
    cfg.DATA_CONFIG.TTA = False
    plain = type(ds)(cfg.DATA_CONFIG, cfg.CLASS_NAMES, root_path=root)
    ob, osc, ol, oc = tta.TTAPipeline(pipe, ds.test_time_augmentor)(plain[0]['points'])
    k = int(oc[0].item())
    assert k == got_b.shape[0]
    np.testing.assert_allclose(got_b.cpu().numpy(), ob[0, :k].cpu().numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(got_s.cpu().numpy(), osc[0, :k].cpu().numpy(), rtol=0, atol=1e-6)
    assert torch.equal(got_l.cpu(), ol[0, :k].long().cpu())
