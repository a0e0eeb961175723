"""BASELINE.json configs[2] and configs[4] as one chain on the device (world size 1 here; the sharding / gather of the same call is
covered by tests/test_frame_parallel.py on gloo): a synthetic SEQUENCE of two-sweep frames (6 point features) ->
multi-sweep detector with the split-precision sparse backbone (DynamicMeanVFE, centerpoint_3sweeps configuration) driven
through ``run_frame_parallel`` in batches of 8 frames, sampler order kept -> result records -> ``prepare_tracker_input``
(tracking/detzero_track/datasets/data_processor.py queue on the HIP overlap kernel) -> object crops of the frames' points in the
kept boxes -> GRM input features (device draw) -> GeometryTransformer.  An interface test: every piece has its own parity test
against the reference; this one checks that the pieces accept each other's outputs, with the oracle on one frame as anchor."""
import numpy as np
import pytest
import torch

from detzero_amd.config import AttrDict
from detzero_amd.synth import VOXEL_SIZE_02, merge_two_sweeps, synth_state_dict, synth_waymo_frame
from tests.util import cpu_state_dict, make_model, match_boxes, oracle_detect

pytestmark = pytest.mark.gpu
CLASSES = ['Vehicle', 'Pedestrian', 'Cyclist']
N_FRAMES = 20          # not a multiple of the batch: the last call is shorter


def _pose(i):
    a = 0.01 * i
    p = np.eye(4)
    p[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
    p[:3, 3] = [2000.0 + 0.8 * i, -700.0 + 0.05 * i, 12.0]
    return p


def test_sequence_detect_track_input_refine(device):
    from detzero_amd import frame_parallel as fp, object_crop, object_features as of, refine_modules as rm, track_adapter as ta
    from detzero_amd.centerpoint import FramePipeline
    from oracle.voxelize import mask_points_by_range
    from tests.test_refine import GCFG
    model, cfg, info = make_model(VOXEL_SIZE_02, seed=4, sweeps=3)
    sd = cpu_state_dict(model)
    pipe = FramePipeline(model.to(device), info, dynamic=True, math='f16x2')
    host = []
    for i in range(N_FRAMES):
        m = merge_two_sweeps(synth_waymo_frame(100 + i, 40000), synth_waymo_frame(101 + i, 40000))     # (enough points for > 10 confident boxes per frame)
        host.append(m[mask_points_by_range(m, info.point_cloud_range)])
    frames = [torch.from_numpy(f).to(device) for f in host]
    metas = [{'sequence_name': 'seq-a', 'frame_id': i, 'pose': _pose(i)} for i in range(N_FRAMES)]
    calls = []

    def batched(fr):
        calls.append(len(fr))
        return pipe(fr)
    annos = fp.run_frame_parallel(batched, frames, CLASSES, metas=metas, batch=8)
    assert calls == [8, 8, 4] and len(annos) == N_FRAMES and [a['frame_id'] for a in annos] == list(range(N_FRAMES))
    # anchor: frame 3 against the CPU oracle (DynamicMeanVFE path), boxes within the north star's 1e-3
    ref = oracle_detect(sd, host[3], info, dynamic=True)['final'][0]
    nm, worst = match_boxes(ref['pred_boxes'].numpy(), ref['pred_scores'].numpy(), annos[3]['boxes_lidar'], annos[3]['score'], tol=1e-3)
    print('sequence anchor: %d reference boxes, %d matched within 1e-3 (worst %.2e)' % (ref['pred_boxes'].shape[0], nm, worst))
    assert ref['pred_boxes'].shape[0] > 10 and nm >= ref['pred_boxes'].shape[0] - 2, (nm, ref['pred_boxes'].shape[0], worst)
    # tracker input (config-named queue; overlap filter on dz_boxes_overlap_bev)
    cfgs = [AttrDict({'NAME': 'heading_process'}), AttrDict({'NAME': 'low_confidence_box_filter', 'THRESHOLD': 0.1}),
            AttrDict({'NAME': 'overlap_box_filter', 'METHOD': 'max_score', 'CLASS_THRESHOLD': {'Vehicle': 0.3, 'Pedestrian': 0.2, 'Cyclist': 0.2}}),
            AttrDict({'NAME': 'transform_to_global'})]
    tracks_in = ta.prepare_tracker_input(annos, cfgs)
    processed, removed = tracks_in['seq-a']
    assert sorted(processed, key=int) == [str(i) for i in range(N_FRAMES)]          # the reference keys frames by str(frame_id)
    for fid, fr in processed.items():
        assert fr['boxes_global'].shape == fr['boxes_lidar'].shape and len(fr['name']) == fr['boxes_lidar'].shape[0] <= annos[int(fid)]['boxes_lidar'].shape[0]
        assert (np.asarray(fr['score']) >= 0.1).all()
    # a stand-in for the CPU tracker (out of scope): the k-th most confident box of every frame forms "track" k
    n_obj, n_t = 3, 6
    tracks = [{'boxes_global': [], 'score': [], 'pts': [], 'name': 'Vehicle'} for _ in range(n_obj)]
    for f in range(n_t):
        fr = processed[str(f)]
        order = np.argsort(-np.asarray(fr['score']))[:n_obj]
        bg = np.asarray(fr['boxes_global'], dtype=np.float64)[order][:, :7]
        raw6 = np.concatenate([host[f][:, :5], -np.ones((host[f].shape[0], 1), np.float32)], axis=1)       # [x,y,z,i,e,NLZ] as stored on disk
        crops = object_crop.crop_frame_objects(raw6, metas[f]['pose'], bg, 1.1, False, device=device)
        for o in range(n_obj):
            tracks[o]['boxes_global'].append(bg[o]); tracks[o]['score'].append(float(np.asarray(fr['score'])[order][o])); tracks[o]['pts'].append(crops[o])
    for t in tracks:
        t['boxes_global'], t['score'] = np.stack(t['boxes_global']), np.asarray(t['score'])
    packed = of.PackedTracks(tracks, device)
    g_in = of.grm_features(packed, rng=of.DeviceDraw(seed=5))
    grm = rm.GeometryTransformer(GCFG, query_point_dims=11, memory_point_dims=4)
    grm.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in grm.state_dict().items()}, seed=9), strict=True)
    grm = grm.eval().to(device).set_math('f16x2')
    out = grm({k: g_in[k] for k in ('geo_memory_points', 'geo_query_points', 'geo_query_boxes', 'geo_query_num')})['batch_box_preds']
    assert out.shape[0] == n_obj and torch.isfinite(out).all()
