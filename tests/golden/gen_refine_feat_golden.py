"""Golden vectors for the object-feature encoding (cropped object points -> GRM / PRM model inputs) from the
REFERENCE's own dataset classes, CPU.

    python tests/golden/gen_refine_feat_golden.py      (build container only; needs /root/reference)

Runs WaymoGeometryDataset.extract_track_feature (refining/detzero_refine/datasets/waymo/waymo_geometry_dataset.py:26-155)
and WaymoPositionDataset.extract_track_feature (waymo_position_dataset.py:31-184) in inference mode, then
DatasetTemplate.collate_batch (datasets/dataset.py:207-258), on seeded synthetic tracks
(detzero_amd.synth.synth_object_track).  The dataset constructors read files; they are bypassed with
object.__new__ and the attributes extract_track_feature uses are set by hand to the values of
ref_dataset_cfgs/waymo_{grm,prm}_dataset.yaml.  Stubbed: the CUDA-extension import of detzero_utils/box_utils.py
(roiaware_pool3d_utils, unused here).  `np.int` (removed from numpy) is aliased to int for data_utils.py:23.
Python's `random` is seeded per object: sample_points draws its subsets from it, and the tests re-draw the same ones.
"""
import importlib
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)

GRM_TRACKS = [(11, 40, 'Vehicle', 0, 300), (12, 2, 'Pedestrian', 10, 40), (13, 17, 'Cyclist', 200, 700), (14, 1, 'Vehicle', 5, 5)]
PRM_TRACKS = [(21, 30, 'Vehicle', 0, 400), (22, 3, 'Pedestrian', 10, 60), (23, 200, 'Cyclist', 0, 90)]
PRM_CLASS_TRACKS = [(31, 9, 'Cyclist', 20, 300)]


def _pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    return m


def import_reference():
    if not hasattr(np, 'int'):
        np.int = int
    _pkg('detzero_utils', REF + '/utils/detzero_utils')
    _pkg('detzero_utils.ops'); _pkg('detzero_utils.ops.roiaware_pool3d')
    stub = types.ModuleType('detzero_utils.ops.roiaware_pool3d.roiaware_pool3d_utils')
    sys.modules[stub.__name__] = stub
    sys.modules['detzero_utils.ops.roiaware_pool3d'].roiaware_pool3d_utils = stub
    _pkg('detzero_refine', REF + '/refining/detzero_refine')
    _pkg('detzero_refine.utils', REF + '/refining/detzero_refine/utils')
    _pkg('detzero_refine.datasets', REF + '/refining/detzero_refine/datasets')
    _pkg('detzero_refine.datasets.waymo', REF + '/refining/detzero_refine/datasets/waymo')
    geo = importlib.import_module('detzero_refine.datasets.waymo.waymo_geometry_dataset')
    pos = importlib.import_module('detzero_refine.datasets.waymo.waymo_position_dataset')
    base = importlib.import_module('detzero_refine.datasets.dataset')
    return geo.WaymoGeometryDataset, pos.WaymoPositionDataset, base.DatasetTemplate


def bare(cls, **attrs):
    ds = object.__new__(cls)
    ds.training = False
    ds.augment_single = ds.augment_full = False
    ds.class_map = {'Vehicle': 1, 'Pedestrian': 2, 'Cyclist': 3, 1: 'Vehicle', 2: 'Pedestrian', 3: 'Cyclist'}
    for k, v in attrs.items():
        setattr(ds, k, v)
    return ds


def data_info(track):
    t = len(track['score'])
    return {'boxes_global': track['boxes_global'].copy(), 'score': track['score'].copy(), 'sample_idx': np.arange(t),
            'pose': np.tile(np.eye(4), (t, 1, 1)), 'pts': [p.copy() for p in track['pts']], 'matched': np.ones(t, dtype=bool),
            'matched_tracklet': True, 'state': 'dynamic', 'gt_boxes_global': track['boxes_global'].copy(),
            'sequence_name': 'synthetic', 'obj_id': 'obj', 'name': track['name']}


def main():
    from detzero_amd.synth import synth_object_track
    Geo, Pos, Base = import_reference()
    out = {}
    # ---- GRM (waymo_grm_dataset.yaml: QUERY_NUM 3, QUERY_POINTS_NUM 256, MEMORY_POINTS_NUM 4096)
    items = []
    for seed, n, name, lo, hi in GRM_TRACKS:
        ds = bare(Geo, query_num=3, query_pts_num=256, memory_pts_num=4096, encoding=['xyz', 'intensity', 'p2s', 'score'])
        random.seed(1000 + seed)
        items.append(ds.extract_track_feature(data_info(synth_object_track(seed, n, name, lo, hi))))
    batch = Base.collate_batch(items)
    out['grm_query_num'] = np.asarray(batch['geo_query_num'])
    for k in ('geo_query_points', 'geo_query_boxes', 'geo_memory_points'):
        out['grm_' + k] = np.asarray(batch[k], dtype=np.float64)
    # ---- PRM (waymo_prm_dataset.yaml: QUERY_NUM 200, QUERY_POINTS_NUM 256, MEMORY_POINTS_NUM 48)
    for tag, tracks, enc in (('prm', PRM_TRACKS, ['xyz', 'intensity', 'p2co', 'score']),
                             ('prmc', PRM_CLASS_TRACKS, ['xyz', 'intensity', 'p2co', 'score', 'class'])):
        items = []
        for seed, n, name, lo, hi in tracks:
            ds = bare(Pos, query_num=200, query_pts_num=256, memory_pts_num=48, encoding=enc)
            random.seed(2000 + seed)
            items.append(ds.extract_track_feature(data_info(synth_object_track(seed, n, name, lo, hi))))
        batch = Base.collate_batch(items)
        for k in ('pos_trajectory', 'pos_init_box', 'padding_mask', 'pos_query_points', 'pos_memory_points'):
            v = np.asarray(batch[k], dtype=np.float64)
            # the padded boxes and points are zeros: keep the fixture small by storing float32 (values are compared at 1e-5)
            out[tag + '_' + k] = v.astype(np.float32) if v.size > 100000 else v
    np.savez_compressed(os.path.join(HERE, 'refine_feat_golden.npz'), **out)
    print('saved %d arrays, %d KiB' % (len(out), os.path.getsize(os.path.join(HERE, 'refine_feat_golden.npz')) // 1024))


if __name__ == '__main__':
    main()
