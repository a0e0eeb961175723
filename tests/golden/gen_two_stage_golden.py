"""Golden vectors of the WHOLE two-stage detector at bench size (BASELINE configs[4] shape: one merged 2-sweep frame of 320 000
points, 0.1 m voxels, centerpoint_pdv_3sweeps-shaped model): first stage on the CPU oracle, second stage on the REFERENCE's own
PDVHead class (imported from /root/reference, its three CUDA extensions replaced by the numpy kernels of oracle/pdv.py exactly as
gen_pdv_golden.py does).

    python tests/golden/gen_two_stage_golden.py     (build container only; needs /root/reference; ~10 min of CPU)
        ->  tests/golden/two_stage_golden.npz

What the chain is: points -> DynamicMeanVFE -> VoxelResBackBone8x -> BEV backbone -> CenterHead -> decode + NMS (oracle/, each
function cites the reference file it restates) -> the first stage's boxes are the RoIs, x_conv3 / x_conv4 of the oracle's backbone
the multi-scale features -> pdv_head.PDVHead.forward (reference code) -> batch_box_preds / batch_cls_preds.
Everything is a function of seeds (frame: synth_waymo_frame(60 / 70); weights: detzero_amd.centerpoint.synth_detector(seed=0,
second_stage=True) - the variance-preserving set, RoIs on occupied cells), so the fixture stores outputs only: the RoIs the second stage saw, its boxes and confidences,
and per-RoI ball-index checksums (to tell a boundary centroid that changed sides from an error).
tests/test_pdv.py::test_two_stage_boxes_at_bench_size runs FramePipeline.two_stage on the same frame on the GPU and compares."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

POINTS_PER_SWEEP = 160000
SEEDS = (60, 70)


def build_model():
    """The model of tools/bench_pdv.py: `synth_detector(second_stage=True)` - the centerpoint_pdv_3sweeps shape on the variance-preserving
    weight set (round 6), so the first stage's boxes - the RoIs of the second stage - sit on the frame's points."""
    from detzero_amd.centerpoint import synth_detector
    return synth_detector((0.1, 0.1, 0.15), seed=0, second_stage=True)


def frame():
    from detzero_amd.synth import merge_two_sweeps, synth_waymo_frame
    return merge_two_sweeps(synth_waymo_frame(SEEDS[0], POINTS_PER_SWEEP), synth_waymo_frame(SEEDS[1], POINTS_PER_SWEEP))


def post_cfg(cfg):
    p = cfg.MODEL.DENSE_HEAD.POST_PROCESSING
    n = p.NMS_CONFIG
    return {'SCORE_THRESH': p.SCORE_THRESH, 'POST_CENTER_LIMIT_RANGE': list(p.POST_CENTER_LIMIT_RANGE), 'MAX_OBJ_PER_SAMPLE': p.MAX_OBJ_PER_SAMPLE,
            'NMS_THRESH': n.NMS_THRESH, 'NMS_PRE_MAXSIZE': n.NMS_PRE_MAXSIZE, 'NMS_POST_MAXSIZE': n.NMS_POST_MAXSIZE}


def main():
    import gen_pdv_golden as gp
    from tests.util import cpu_state_dict, oracle_detect
    model, cfg, info = build_model()
    sd = cpu_state_dict(model)
    pts = frame()
    t0 = time.time()
    ref = oracle_detect(sd, pts, info, post=post_cfg(cfg), dynamic=True)
    fin = ref['final'][0]
    k = int(fin['pred_boxes'].shape[0])
    print('oracle first stage: %d boxes in %.0f s' % (k, time.time() - t0))
    assert k > 50
    res = ref['backbone']
    f3, c3, s3 = res['x_conv3']
    f4, c4, s4 = res['x_conv4']
    pdv = gp.install_reference()
    from detzero_amd.config import AttrDict
    head = pdv.PDVHead(model.backbone2d.num_bev_features, AttrDict(cfg.MODEL.ROI_HEAD), info.point_cloud_range, info.voxel_size, num_class=1).eval()
    head.load_state_dict({kk[len('roi_head.'):]: v for kk, v in sd.items() if kk.startswith('roi_head.')}, strict=True)
    t = torch.from_numpy
    pb = np.concatenate([np.zeros((pts.shape[0], 1), np.float32), pts], 1)
    rois = fin['pred_boxes'].numpy().astype(np.float32)[None]
    scores = fin['pred_scores'].numpy().astype(np.float32)[None]
    labels = fin['pred_labels'].numpy().astype(np.int64)[None]
    bd = {'batch_size': 1, 'points': t(pb), 'rois': t(rois), 'roi_scores': t(scores), 'roi_labels': t(labels), 'has_class_labels': True,
          'multi_scale_3d_strides': {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8},
          'multi_scale_3d_features': {'x_conv3': gp.SparseStub(t(np.ascontiguousarray(c3)).int(), f3 if torch.is_tensor(f3) else t(f3), s3, 1),
                                      'x_conv4': gp.SparseStub(t(np.ascontiguousarray(c4)).int(), f4 if torch.is_tensor(f4) else t(f4), s4, 1)}}
    t0 = time.time()
    with torch.no_grad():
        pf, pc = head.get_point_voxel_features(bd)
        bd['point_features'], bd['point_coords'] = pf, pc
        _, _, _, ball = head.roi_grid_pool(bd)
        out = head({kk: v for kk, v in bd.items() if kk not in ('point_features', 'point_coords')})
    print('reference PDVHead on %d points, %d RoIs: %.0f s' % (pts.shape[0], k, time.time() - t0))
    small = {'n_points': np.array(pts.shape[0]), 'n_c3': np.array(c3.shape[0]), 'n_c4': np.array(c4.shape[0]), 'rois': rois[0], 'roi_scores': scores[0],
             'roi_labels': labels[0], 'ball_row_sums': ball.numpy().astype(np.int64).sum(axis=(1, 2)),
             'batch_box_preds': out['batch_box_preds'].numpy()[0].astype(np.float32), 'batch_cls_preds': out['batch_cls_preds'].numpy()[0].astype(np.float32)}
    print('RoIs with at least one non-empty ball: %d of %d' % (int((small['ball_row_sums'] != 0).sum()), k))
    path = os.path.join(HERE, 'two_stage_golden.npz')
    np.savez_compressed(path, **small)
    print('saved', os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
