"""Golden vectors for test-time augmentation (point transforms, box restore, weighted box fusion) from the REFERENCE's
own code, CPU.

    python tests/golden/gen_tta_golden.py      (build container only; needs /root/reference)

Runs, unmodified: TestTimeAugmentor (detection/detzero_det/datasets/augmentor/test_time_augmentor.py) with the
TEST_TIME_AUGMENTOR block of det_dataset_cfgs/waymo_1sweep.yaml:48-58; the source of CenterPoint.test_time_augment
(detection/detzero_det/models/centerpoint.py:131-208, extracted from the file with `ast` because importing the module pulls
in spconv); wbf_online (utils/ensemble_utils/ensemble.py:7-33) -> weighted_boxes_fusion_3d (wbf_3d.py:118-203) ->
boxes_iou3d_gpu (utils/detzero_utils/ops/iou3d_nms/iou3d_nms_utils.py:74-107).  Replaced: the CUDA extension call
iou3d_nms_cuda.boxes_overlap_bev_gpu, by the oracle's rotated overlap (oracle/c/oracle.c, itself pinned bit-exact
against the reference's iou3d_cpu.cpp); `.cuda()` / torch.cuda.FloatTensor, by their CPU twins.
Inputs: seeded synthetic detections of one frame under every augmentation (clusters of agreeing boxes, misses, false
positives), built with the forward box transforms below.
"""
import ast
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)

AUG_CONFIG = [{'NAME': 'world_flip', 'ALONG_AXIS_LIST': ['x', 'y', 'xy']},
              {'NAME': 'world_rotation', 'ROT_ANGLE': [0, -0.39365818, -0.78539816, -1.17809724, -2.74889357, 0.39365818, 0.78539816,
                                                       1.17809724, 2.74889357, 3.14159265]},
              {'NAME': 'world_scaling', 'SCALE_RANGE': [0.95, 1.05]}]
FRAMES = [(3, 70), (4, 25), (5, 2)]           # (seed, objects) of the fusion cases (the reference itself fails on "no detection in any copy": reshape of 0 elements)


def _mod(name, path=None, **attrs):
    m = types.ModuleType(name)
    if path:
        m.__path__ = [path]
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def import_reference():
    from oracle import cref
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor

    def boxes_overlap_bev_gpu(a, b, out):
        out.copy_(torch.from_numpy(cref.boxes_overlap_bev(a.numpy().astype(np.float32), b.numpy().astype(np.float32)).astype(np.float32)))
        return 1
    _mod('detzero_utils', REF + '/utils/detzero_utils')
    cu = _load('detzero_utils.common_utils', REF + '/utils/detzero_utils/common_utils.py')
    sys.modules['detzero_utils'].common_utils = cu
    _mod('detzero_utils.ops'); _mod('detzero_utils.ops.iou3d_nms')
    ext = _mod('detzero_utils.ops.iou3d_nms.iou3d_nms_cuda', boxes_overlap_bev_gpu=boxes_overlap_bev_gpu)
    sys.modules['detzero_utils.ops.iou3d_nms'].iou3d_nms_cuda = ext
    _load('detzero_utils.ops.iou3d_nms.iou3d_nms_utils', REF + '/utils/detzero_utils/ops/iou3d_nms/iou3d_nms_utils.py')
    _mod('detzero_det'); _mod('detzero_det.utils')
    _mod('detzero_det.utils.ensemble_utils', REF + '/detection/detzero_det/utils/ensemble_utils')
    ens = importlib.import_module('detzero_det.utils.ensemble_utils.ensemble')
    aug = _load('ref_tta', REF + '/detection/detzero_det/datasets/augmentor/test_time_augmentor.py')
    # CenterPoint.test_time_augment as a free function
    src = open(REF + '/detection/detzero_det/models/centerpoint.py').read()
    fn = None
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.FunctionDef) and node.name == 'test_time_augment':
            node.decorator_list = []
            fn = node
    code = compile(ast.Module(body=[fn], type_ignores=[]), 'centerpoint.py:test_time_augment', 'exec')
    captured = {}

    def wbf_online(boxes, scores, labels):
        captured['restored'] = boxes.clone()
        return ens.wbf_online(boxes, scores, labels)
    glb = {'torch': torch, 'np': np, 'common_utils': cu, 'wbf_online': wbf_online}
    exec(code, glb)
    captured['wbf_tracking_v1'] = ens.wbf_tracking_v1
    return aug.TestTimeAugmentor, glb['test_time_augment'], captured


def forward_boxes(boxes, op):
    """What an object at `boxes` (original frame) looks like in the augmented copy `op` (inverse of the restore)."""
    b = boxes.astype(np.float64).copy()
    if op == 'tta_original':
        return b
    _, name, param = op.split('_')
    if name == 'flip':
        if param == 'x':
            b[:, 1], b[:, 6] = -b[:, 1], -b[:, 6]
        elif param == 'y':
            b[:, 0], b[:, 6] = -b[:, 0], -b[:, 6] - np.pi
        else:
            b[:, 0:2], b[:, 6] = -b[:, 0:2], b[:, 6] - np.pi
    elif name == 'rot':
        a = float(param)
        c, s = np.cos(a), np.sin(a)
        b[:, 0], b[:, 1] = boxes[:, 0] * c - boxes[:, 1] * s, boxes[:, 0] * s + boxes[:, 1] * c
        b[:, 6] += a
    elif name == 'scale':
        b[:, :6] *= float(param)
    return b


def synth_predictions(seed, n_obj, ops):
    """Per augmented copy: the objects it found (most of them, jittered), its false positives; distinct scores."""
    from detzero_amd.synth import synth_boxes
    rng = np.random.default_rng(seed)
    gt = synth_boxes(seed, max(n_obj, 1), xy_range=60.0, near_duplicates=0.0).astype(np.float64)[:n_obj]
    cls = rng.integers(1, 4, size=n_obj)
    base = rng.uniform(0.05, 0.95, size=n_obj)
    preds = []
    seen = set()
    for op in ops:
        keep = rng.random(n_obj) < 0.85
        b = gt[keep].copy()
        b[:, :3] += rng.normal(0, 0.04, size=(b.shape[0], 3))
        b[:, 3:6] *= rng.uniform(0.97, 1.03, size=(b.shape[0], 3))
        b[:, 6] += rng.normal(0, 0.02, size=b.shape[0])
        sc = np.clip(base[keep] + rng.normal(0, 0.05, size=b.shape[0]), 0.005, 0.999) * rng.uniform(0.98, 1.0, size=b.shape[0])
        lab = cls[keep]
        n_fp = int(rng.integers(0, 6)) if n_obj else 0
        fp = synth_boxes(seed * 100 + len(preds), max(n_fp, 1), xy_range=60.0, near_duplicates=0.0).astype(np.float64)[:n_fp]
        b = np.concatenate([b, fp], axis=0)
        sc = np.concatenate([sc, rng.uniform(0.005, 0.3, size=n_fp)])
        lab = np.concatenate([lab, rng.integers(1, 4, size=n_fp)])
        # no two candidates of a frame share a score: the reference orders candidates with argsort()[::-1], whose order
        # among equal keys is numpy's implementation detail (and decides which member lends its heading to a cluster)
        sc = sc.astype(np.float32)
        for i in range(sc.shape[0]):
            while float(sc[i]) in seen:
                sc[i] = np.nextafter(sc[i], np.float32(0))
            seen.add(float(sc[i]))
        perm = np.argsort(-sc)
        preds.append({'pred_boxes': torch.from_numpy(forward_boxes(b[perm], op).astype(np.float32)),
                      'pred_scores': torch.from_numpy(sc[perm].astype(np.float32)),
                      'pred_labels': torch.from_numpy(lab[perm].astype(np.int64))})
    return preds


def main():
    from detzero_amd.config import AttrDict
    from detzero_amd.synth import synth_waymo_frame
    Aug, test_time_augment, captured = import_reference()
    out = {}
    aug = Aug([AttrDict(c) for c in AUG_CONFIG])
    pts = synth_waymo_frame(17, n_points=4096)
    copies = aug.forward({'points': pts.copy(), 'frame_id': 0})
    ops = list(copies.keys())
    out['ops'] = np.array(ops)
    out['points'] = pts
    for k, v in copies.items():
        out['points_' + k] = np.asarray(v['points'], dtype=np.float32)
    for seed, n_obj in FRAMES:
        preds = synth_predictions(seed, n_obj, ops)
        m = max(len(p['pred_boxes']) for p in preds)
        tag = 'f%d_' % seed
        pb, ps, pl = np.zeros((len(ops), m, 7), np.float32), np.zeros((len(ops), m), np.float32), np.zeros((len(ops), m), np.int32)
        for i, p in enumerate(preds):
            n = len(p['pred_boxes'])
            pb[i, :n], ps[i, :n], pl[i, :n] = p['pred_boxes'].numpy(), p['pred_scores'].numpy(), p['pred_labels'].numpy()
        out[tag + 'pred_boxes'], out[tag + 'pred_scores'], out[tag + 'pred_labels'] = pb, ps, pl
        boxes, scores, labels = test_time_augment({'tta_ops': ops, 'batch_size': len(ops)}, preds)
        out[tag + 'restored'] = captured['restored'].numpy()
        out[tag + 'fused_boxes'] = boxes.numpy()
        out[tag + 'fused_scores'] = scores.numpy()
        out[tag + 'fused_labels'] = labels.numpy()
        print(tag, 'copies', len(ops), 'max boxes', m, '->', boxes.shape[0], 'fused')
        if seed == FRAMES[1][0]:
            # tracking variant (ensemble.py:35-62 -> wbf_3d.py:205-265) on the same restored boxes: object ids per candidate,
            # -1 (no id) for a third of them
            rng = np.random.default_rng(seed)
            ids = np.where(rng.random(pl.shape) < 0.33, -1, rng.integers(0, 40, size=pl.shape)).astype(np.int64)
            out[tag + 'obj_ids'] = ids
            tb, ts, tl, ti = captured['wbf_tracking_v1'](captured['restored'].clone(), torch.from_numpy(ps)[..., None], torch.from_numpy(pl)[..., None],
                                                         torch.from_numpy(ids)[..., None])
            out[tag + 'trk_boxes'], out[tag + 'trk_scores'], out[tag + 'trk_labels'], out[tag + 'trk_ids'] = tb.numpy(), ts.numpy(), tl.numpy(), ti.numpy()
            print(tag, 'tracking variant ->', tb.shape[0], 'fused,', int((ti >= 0).sum()), 'with an id')
    np.savez_compressed(os.path.join(HERE, 'tta_golden.npz'), **out)
    print('saved %d arrays, %d KiB' % (len(out), os.path.getsize(os.path.join(HERE, 'tta_golden.npz')) // 1024))


if __name__ == '__main__':
    main()
