"""Golden vectors for row C5 (recall record) from the REFERENCE's own code, CPU.

    python tests/golden/gen_recall_golden.py      (build container only; needs /root/reference)

CenterPoint.generate_recall_record (detection/detzero_det/models/centerpoint.py:310-352) and iou3d_nms_utils.boxes_iou3d_gpu
(utils/detzero_utils/ops/iou3d_nms/iou3d_nms_utils.py:74-107) are extracted with `ast` and executed as they are; the only stand-ins
are the CUDA extension call `iou3d_nms_cuda.boxes_overlap_bev_gpu` (the oracle's overlap routine, pinned bit-exact on the
reference's iou3d_cpu.cpp - tests/test_oracle_golden.py) and `torch.cuda.FloatTensor` (the CPU constructor).
Scenes: padded ground truth, predictions that are perturbed / shifted / unrelated boxes, an empty prediction list, `rois`.
"""
import ast
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
THRESH = [0.3, 0.5, 0.7]


def _extract(path, name):
    for node in ast.walk(ast.parse(open(path).read())):
        if isinstance(node, ast.FunctionDef) and node.name == name:
            node.decorator_list = []
            return node
    raise KeyError(name)


def reference_functions():
    from oracle import cref

    def overlap_stub(a, b, out):
        out.copy_(torch.from_numpy(cref.boxes_overlap_bev(a.numpy().astype(np.float32), b.numpy().astype(np.float32))))
        return 1
    tp = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith('__')})
    tp.cuda = types.SimpleNamespace(FloatTensor=torch.FloatTensor)
    g1 = {'torch': tp, 'iou3d_nms_cuda': types.SimpleNamespace(boxes_overlap_bev_gpu=overlap_stub)}
    node = _extract(REF + '/utils/detzero_utils/ops/iou3d_nms/iou3d_nms_utils.py', 'boxes_iou3d_gpu')
    exec(compile(ast.Module(body=[node], type_ignores=[]), 'iou3d_nms_utils.py:boxes_iou3d_gpu', 'exec'), g1)
    g2 = {'torch': torch, 'iou3d_nms_utils': types.SimpleNamespace(boxes_iou3d_gpu=g1['boxes_iou3d_gpu'])}
    node = _extract(REF + '/detection/detzero_det/models/centerpoint.py', 'generate_recall_record')
    exec(compile(ast.Module(body=[node], type_ignores=[]), 'centerpoint.py:generate_recall_record', 'exec'), g2)
    return g2['generate_recall_record'], g1['boxes_iou3d_gpu']


def scene(seed, n_gt, n_pad, n_noise, pseed=0):
    from detzero_amd.synth import synth_boxes
    rng = np.random.default_rng(seed + 7919 * pseed)
    gt = synth_boxes(seed, n_gt, near_duplicates=0.0).astype(np.float32)
    pred = gt.copy()
    kind = rng.integers(0, 4, size=n_gt)                                   # 0 tight, 1 loose, 2 far, 3 missing
    pred[:, :3] += (rng.normal(0, 1, (n_gt, 3)) * np.array([0.05, 0.05, 0.02])).astype(np.float32)
    loose = kind == 1
    pred[loose, 0] += (0.3 * gt[loose, 3] * rng.choice([-1, 1], loose.sum())).astype(np.float32)
    pred[loose, 6] += rng.normal(0, 0.15, loose.sum()).astype(np.float32)
    far = kind == 2
    pred[far, :2] += (1.5 * gt[far, 3:5]).astype(np.float32)
    pred = pred[kind != 3]
    noise = synth_boxes(seed + 999, max(n_noise, 1), near_duplicates=0.0).astype(np.float32)[:n_noise]
    pred = np.concatenate([pred, noise], 0)[rng.permutation(pred.shape[0] + n_noise)]
    labels = rng.integers(1, 4, size=(n_gt, 1)).astype(np.float32)
    gt_pad = np.concatenate([np.concatenate([gt, labels], 1), np.zeros((n_pad, 8), np.float32)], 0)
    return pred.astype(np.float32), gt_pad


def main():
    rec_fn, iou_fn = reference_functions()
    out = {'thresh': np.array(THRESH, np.float32)}
    recall = {}
    cases = [(3, 24, 5, 6), (4, 40, 0, 10), (5, 12, 7, 0), (6, 16, 3, 4)]
    for ci, (seed, n_gt, n_pad, n_noise) in enumerate(cases):
        pred, gt_pad = scene(seed, n_gt, n_pad, n_noise)
        if ci == 2:
            pred = pred[:0]                                                # a frame without detections
        data = {'gt_boxes': torch.from_numpy(gt_pad)[None]}
        if ci == 3:
            data['rois'] = torch.from_numpy(scene(seed, n_gt, 0, 2, pseed=1)[0])[None]      # a second-stage record: rois (another perturbation of the same objects) + refined boxes
        for boxes in ([pred] if pred.shape[0] else []) + ([data['rois'][0].numpy()] if 'rois' in data else []):
            iou = iou_fn(torch.from_numpy(boxes[:, :7]), torch.from_numpy(gt_pad[:n_gt, :7])).max(dim=0)[0].numpy()
            assert not np.any(np.abs(iou[:, None] - np.array(THRESH)[None, :]) < 2e-3), 'a ground-truth box sits on a threshold'
        recall = rec_fn(torch.from_numpy(pred), recall, 0, data, THRESH)
        out['pred%d' % ci] = pred
        out['gt%d' % ci] = gt_pad
        if 'rois' in data:
            out['rois%d' % ci] = data['rois'][0].numpy()
        out['recall_after%d' % ci] = np.array([recall['gt']] + [recall['roi_%s' % t] for t in THRESH] + [recall['rcnn_%s' % t] for t in THRESH], np.int64)
        print(ci, dict(recall))
    out['n_cases'] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, 'recall_golden.npz'), **out)


if __name__ == '__main__':
    main()
