"""The refining data path end to end on the device: tracked boxes + raw frames -> per-object crops -> GRM / PRM / CRM inputs
(device-side draw) -> the three models -> per-frame result records.  An interface test (the pieces have their own parity
tests): shapes, keys, finiteness and the geometric consistency of what comes back."""
import numpy as np
import pytest
import torch

from detzero_amd.config import AttrDict
from detzero_amd.synth import synth_state_dict, synth_waymo_frame


@pytest.mark.gpu
def test_crop_features_models_write_back(device):
    from detzero_amd import object_crop, object_features as of, refine_modules as rm, refine_results as rr
    from detzero_amd.track_adapter import transform_boxes3d
    from tests.test_refine import GCFG, PCFG
    rng = np.random.default_rng(3)
    n_frames, n_obj = 6, 5
    # a short sequence: ego poses, raw 6-column frames, and tracked boxes (lidar frame) of a few objects seen in every frame
    poses, frames, boxes_l = [], [], []
    base = np.stack([rng.uniform(-25, 25, n_obj), rng.uniform(-25, 25, n_obj), rng.uniform(0.5, 1.0, n_obj), rng.uniform(3, 5, n_obj),
                     rng.uniform(1.6, 2.2, n_obj), rng.uniform(1.4, 1.9, n_obj), rng.uniform(-3, 3, n_obj)], axis=1)
    for f in range(n_frames):
        a = 0.02 * f
        pose = np.eye(4)
        pose[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        pose[:3, 3] = [1000 + 1.0 * f, -500 + 0.1 * f, 10]
        poses.append(pose)
        p5 = synth_waymo_frame(40 + f, 60000)
        frames.append(np.concatenate([p5, -np.ones((p5.shape[0], 1), np.float32)], axis=1))
        b = base.copy()
        b[:, 0] += 0.3 * f
        boxes_l.append(b)
    tracks = [{'boxes_global': [], 'score': [], 'pts': [], 'name': 'Vehicle'} for _ in range(n_obj)]
    for f in range(n_frames):
        bg = transform_boxes3d(boxes_l[f].copy(), poses[f])
        crops = object_crop.crop_frame_objects(frames[f], poses[f], bg, 1.1, False, device=device)
        for o in range(n_obj):
            tracks[o]['boxes_global'].append(bg[o]); tracks[o]['score'].append(0.3 + 0.1 * o + 0.01 * f); tracks[o]['pts'].append(crops[o])
    for t in tracks:
        t['boxes_global'], t['score'] = np.stack(t['boxes_global']), np.asarray(t['score'])
    assert sum(p.shape[0] for t in tracks for p in t['pts']) > 0
    packed = of.PackedTracks(tracks, device)
    dd = of.DeviceDraw(seed=11)
    meta = {'sequence_name': ['seq'] * n_obj, 'obj_id': list(range(n_obj)), 'frame': [np.arange(n_frames)] * n_obj, 'obj_cls': [1] * n_obj,
            'pose': [np.stack(poses)] * n_obj, 'state': ['dynamic'] * n_obj, 'geo_trajectory': [t['boxes_global'] for t in tracks],
            'geo_score': [t['score'] for t in tracks], 'pos_scores': [t['score'] for t in tracks]}

    def load(model, seed):
        model.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=seed), strict=True)
        return model.eval().to(device).set_math('f16x2')
    grm = load(rm.GeometryTransformer(GCFG, query_point_dims=11, memory_point_dims=4), 5)
    prm = load(rm.PositionTransformer(PCFG, query_point_dims=32, memory_point_dims=32), 6)
    crm = load(rm.ConfidencePointnet(AttrDict({'ENCODER_MLP': [128, 128], 'REGRESSION_MLP': [512]}), query_point_dims=32, memory_point_dims=32), 7)

    g_in = of.grm_features(packed, rng=dd)
    g_out = grm({k: g_in[k] for k in ('geo_memory_points', 'geo_query_points', 'geo_query_boxes', 'geo_query_num')})['batch_box_preds']
    res = rr.grm_prediction_dicts(meta, g_out[:, 0] if g_out.dim() == 3 else g_out)
    rec = res['seq'][2]
    assert len(rec['boxes_lidar']) == n_frames and np.isfinite(np.asarray(rec['boxes_lidar'])).all()
    # the GRM only replaces the size: centres come back to where the tracker had them in each frame's lidar coordinates
    np.testing.assert_allclose(np.asarray(rec['boxes_lidar'])[:, 0, :3], np.stack([b[2, :3] for b in boxes_l]), atol=1e-6)

    p_in = of.prm_features(packed, rng=dd)
    p_out = prm({k: p_in[k] for k in ('pos_query_points', 'pos_memory_points', 'pos_trajectory', 'padding_mask')})['batch_box_preds']
    meta['pos_init_box'] = p_in['pos_init_box']
    res = rr.prm_prediction_dicts(meta, p_out)
    rec = res['seq'][4]
    assert np.asarray(rec['boxes_lidar']).shape == (n_frames, 7) and np.isfinite(np.asarray(rec['boxes_global'])).all()
    # feeding the input trajectory itself through the write-back returns the tracked boxes (round trip of the two frames)
    lidar, world = rr.prm_revert_to_each_frame(p_in['pos_trajectory'].double(), p_in['pos_init_box'], meta['pose'])
    np.testing.assert_allclose(world[4][:, :3], tracks[4]['boxes_global'][:, :3], atol=5e-3)
    np.testing.assert_allclose(lidar[4][:, :3], np.stack([b[4, :3] for b in boxes_l]), atol=5e-3)

    c_in = of.crm_features(packed, rng=dd)
    score = crm({'conf_points': c_in['conf_points']})['pred_score']
    meta.update(conf_score=c_in['conf_score'], box_num=c_in['box_num'])
    res = rr.crm_prediction_dicts(meta, score)
    assert res['seq'][0]['new_score'].shape == (n_frames,) and ((res['seq'][0]['new_score'] >= 0) & (res['seq'][0]['new_score'] <= 1)).all()
