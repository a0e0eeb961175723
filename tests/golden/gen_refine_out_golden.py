"""Golden vectors for the refiner's write-back (object-frame predictions -> per-frame boxes / result records) from the
REFERENCE's own dataset classes, CPU.

    python tests/golden/gen_refine_out_golden.py      (build container only; needs /root/reference)

WaymoGeometryDataset.generate_prediction_dicts / revert_to_each_frame (waymo_geometry_dataset.py:160-250),
WaymoPositionDataset.generate_prediction_dicts / revert_to_each_frame (waymo_position_dataset.py:190-286) and
WaymoConfidenceDataset.generate_prediction_dicts (waymo_confidence_dataset.py:164-196) on seeded synthetic batches;
constructors bypassed as in gen_refine_feat_golden.py.  `np.int` (removed from numpy) aliased to int.
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

OBJECTS = [(61, 12, 1), (62, 3, 2), (63, 40, 3)]       # (seed, frames, class id)


def synth_batch():
    """What the collated batch + model outputs look like at write-back time (numpy on the host)."""
    from detzero_amd.synth import synth_object_track
    import gen_tta_golden  # noqa: F401  (not used; keeps the generators' import layout uniform)
    from gen_waymo_io_golden import synth_sequence
    b = {'sequence_name': [], 'obj_id': [], 'frame': [], 'geo_trajectory': [], 'geo_score': [], 'pos_scores': [], 'obj_cls': [], 'pose': [],
         'state': [], 'pos_init_box': [], 'gt_pos_trajectory': [], 'box_num': []}
    rng = np.random.default_rng(5)
    for seed, t, cls in OBJECTS:
        tr = synth_object_track(seed, t, {1: 'Vehicle', 2: 'Pedestrian', 3: 'Cyclist'}[cls], 5, 50)
        infos, _ = synth_sequence(seed, n_frames=t, n_points=64)
        poses = np.stack([i['pose'] for i in infos])
        poses[:, :3, 3] = tr['boxes_global'][:, :3] + rng.uniform(-30, 30, size=(t, 3)) * [1, 1, 0.05]
        b['sequence_name'].append('seq%d' % (seed % 2)); b['obj_id'].append('obj%d' % seed)
        b['frame'].append(np.arange(t) + 10); b['geo_trajectory'].append(tr['boxes_global'].copy())
        b['geo_score'].append(tr['score']); b['pos_scores'].append(tr['score']); b['obj_cls'].append(cls); b['pose'].append(poses)
        b['state'].append('dynamic'); b['box_num'].append(t)
        init = tr['boxes_global'][t // 2].copy()
        init[6] = (init[6] + np.pi) % (2 * np.pi) - np.pi
        b['pos_init_box'].append(init)
        b['gt_pos_trajectory'].append(np.zeros((200, 7)))
    nb = len(OBJECTS)
    grm_pred = np.abs(rng.normal(2.0, 0.8, size=(nb, 7))).astype(np.float32)
    prm_pred = rng.normal(0, 3.0, size=(nb, 200, 7)).astype(np.float32)
    prm_pred[:, :, 3:6] = np.abs(prm_pred[:, :, 3:6]) + 0.5
    crm_pred = rng.uniform(0, 1, size=(nb, 200)).astype(np.float32)
    conf_score = np.full((nb, 200), -1.0)
    for i, (_, t, _) in enumerate(OBJECTS):
        conf_score[i, :t] = b['geo_score'][i]
    b['conf_score'] = conf_score
    return b, grm_pred, prm_pred, crm_pred


def _flatten(tag, res, out):
    for seq in sorted(res):
        for obj in sorted(res[seq]):
            for k, v in res[seq][obj].items():
                if k in ('sequence_name', 'state', 'name'):
                    out['%s_%s_%s_%s' % (tag, seq, obj, k)] = np.asarray(v).astype(str)
                else:
                    out['%s_%s_%s_%s' % (tag, seq, obj, k)] = np.asarray(v)


def main():
    import gen_refine_feat_golden as feat
    Geo, Pos, Base = feat.import_reference()
    conf = importlib.import_module('detzero_refine.datasets.waymo.waymo_confidence_dataset')
    b, grm_pred, prm_pred, crm_pred = synth_batch()
    out = {}
    res = {}
    feat.bare(Geo).generate_prediction_dicts(b, {'pred_boxes': grm_pred, 'geo_trajectory': b['geo_trajectory'], 'pose': b['pose']}, res)
    _flatten('grm', res, out)
    res = {}
    feat.bare(Pos).generate_prediction_dicts(b, {'pred_boxes': prm_pred, 'pos_init_box': b['pos_init_box'], 'pose': b['pose'],
                                                 'gt_pos_trajectory': b['gt_pos_trajectory']}, res)
    for seq in res:
        for obj in res[seq]:
            res[seq][obj].pop('boxes_gt'); res[seq][obj].pop('boxes_gt_global')
    _flatten('prm', res, out)
    res = {}

    class _T:                       # conf_score is a tensor in the reference's batch (`.cpu().numpy()`)
        def __init__(self, a): self.a = a
        def cpu(self): return self
        def numpy(self): return self.a
    b2 = dict(b)
    b2['conf_score'] = _T(b['conf_score'])
    b2['batch_size'] = len(OBJECTS)
    feat.bare(conf.WaymoConfidenceDataset).generate_prediction_dicts(b2, {'pred_score': crm_pred}, res)
    _flatten('crm', res, out)
    np.savez_compressed(os.path.join(HERE, 'refine_out_golden.npz'), **out)
    print('saved %d arrays, %d KiB' % (len(out), os.path.getsize(os.path.join(HERE, 'refine_out_golden.npz')) // 1024))


if __name__ == '__main__':
    main()
