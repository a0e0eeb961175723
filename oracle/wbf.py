"""ORACLE (test infrastructure only): CPU restatement of the test-time-augmentation merge of the detector,

  restore   detection/detzero_det/models/centerpoint.py:131-208 (CenterPoint.test_time_augment: boxes of the augmented
            copies back to the original frame)
  fuse      detection/detzero_det/utils/ensemble_utils/wbf_3d.py:10-203 (prefilter_boxes, get_weighted_box,
            find_matching_box, weighted_boxes_fusion_3d) as called by ensemble.py:7-33 (wbf_online: iou_thr
            [0.8, 0.6, 0.7], skip_box_thr [0.1, 0.01, 0.01], conf_type 'avg', iou_type '3d', allows_overflow False)
  3-D IoU   utils/detzero_utils/ops/iou3d_nms/iou3d_nms_utils.py:74-107, with the rotated BEV overlap of
            oracle/c/oracle.c (pinned bit-exact against the reference's iou3d_cpu.cpp)
  augment   detection/detzero_det/datasets/augmentor/test_time_augmentor.py:32-83 (point-side transforms)

Pinned by tests/golden/tta_golden.npz (gen_tta_golden.py runs the reference's own wbf_3d.py / test_time_augment source
with the CUDA IoU extension replaced by the oracle overlap).  Scalar arithmetic follows the numpy this container has
(NumPy >= 2 promotion: float32 scalar x Python int stays float32), which is what the fixture was produced with.
"""
import numpy as np

from . import cref

IOU_THR = (0.8, 0.6, 0.7)
SKIP_THR = (0.1, 0.01, 0.01)


# ------------------------------------------------------------------------------------------------ point-side augmentation
def rotate_z_f32(points, angle):
    """common_utils.py:220-244 on one float32 array: float32 cos / sin, float32 matmul."""
    p = np.asarray(points, dtype=np.float32)
    a = np.float32(angle)
    c, s = np.cos(a), np.sin(a)
    rot = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]], dtype=np.float32)
    out = p.copy()
    out[:, :3] = p[:, :3] @ rot
    return out


def augment_points(points, op):
    """op: 'tta_original' | 'tta_flip_x|y|xy' | 'tta_rot_<angle>' | 'tta_scale_<factor>' (test_time_augmentor.py:32-83)."""
    p = np.array(points, dtype=np.float32, copy=True)
    if op == 'tta_original':
        return p
    _, name, param = op.split('_')
    if name == 'flip':
        if 'y' in param:
            p[:, 0] = -p[:, 0]
        if 'x' in param:
            p[:, 1] = -p[:, 1]
    elif name == 'rot':
        p = rotate_z_f32(p, float(param))
    elif name == 'scale':
        p[:, :3] *= np.float32(float(param))
    else:
        raise NotImplementedError(op)
    return p


# ------------------------------------------------------------------------------------------------ box-side restore
def restore_boxes(boxes, tta_ops):
    """boxes (tta, max, 7) float32 of ONE frame, rows of augmented copy i in that copy's coordinates -> original frame
    (centerpoint.py:165-203; float32 tensor arithmetic, Python scalars enter as float32)."""
    b = np.array(boxes, dtype=np.float32, copy=True)
    pi = np.float32(np.pi)
    for i, op in enumerate(tta_ops):
        if op == 'tta_original':
            continue
        _, name, param = op.split('_')
        if name == 'flip':
            if param == 'x':
                b[i, :, 1] = -b[i, :, 1]
                b[i, :, 6] = -b[i, :, 6]
            elif param == 'y':
                b[i, :, 0] = -b[i, :, 0]
                b[i, :, 6] = -(b[i, :, 6] + pi)
            elif param == 'xy':
                b[i, :, 0:2] = -b[i, :, 0:2]
                b[i, :, 6] = b[i, :, 6] + pi
        elif name == 'rot':
            ang = -float(param)
            b[i, :, 0:3] = rotate_z_f32(b[i, :, 0:3], ang)[:, :3]
            b[i, :, 6] += np.float32(ang)
        elif name == 'scale':
            b[i, :, :6] /= np.float32(float(param))
    return b


# ------------------------------------------------------------------------------------------------ fusion
def iou3d_one_to_many(box, others):
    """iou3d_nms_utils.py:74-107 for a (7,) box against (C,7) boxes, float32 throughout."""
    a = np.asarray(box, dtype=np.float32).reshape(1, 7)
    b = np.ascontiguousarray(others, dtype=np.float32)
    bev = cref.boxes_overlap_bev(a, b).astype(np.float32)[0]
    a_max, a_min = a[0, 2] + a[0, 5] / np.float32(2), a[0, 2] - a[0, 5] / np.float32(2)
    b_max, b_min = b[:, 2] + b[:, 5] / np.float32(2), b[:, 2] - b[:, 5] / np.float32(2)
    h = np.maximum(np.minimum(a_max, b_max) - np.maximum(a_min, b_min), np.float32(0))
    o3 = bev * h
    vol_a = a[0, 3] * a[0, 4] * a[0, 5]
    vol_b = b[:, 3] * b[:, 4] * b[:, 5]
    return (o3 / np.maximum(vol_a + vol_b - o3, np.float32(1e-6))).astype(np.float32)


def _fused(members, conf_type):
    """get_weighted_box (wbf_3d.py:53-96): rows [label, conf, (obj id,) x, y, z, dx, dy, dz, yaw]; float32 accumulator."""
    out = np.zeros(len(members[0]), dtype=np.float32)
    conf, confs = 0, []
    for m in members:
        out[-7:] += m[1] * m[-7:]
        conf += m[1]
        confs.append(m[1])
    out[0] = members[0][0]
    out[1] = conf / len(members) if conf_type == 'avg' else np.array(confs).max()
    out[-7:] /= conf
    out[-1] = members[confs.index(max(confs))][-1]
    if len(members[0]) > 9:                                                     # tracking rows: id of the most confident member that has one
        ids = np.array([m[2] for m in members])[np.argsort(np.array(confs))[::-1]]
        ids = ids[ids >= 0]
        out[2] = ids[0] if len(ids) else -1
    return out


def weighted_boxes_fusion_3d(boxes, scores, labels, weights=None, iou_thr=IOU_THR, skip_box_thr=SKIP_THR, conf_type='avg',
                             allows_overflow=False, obj_ids=None):
    """boxes (T, M, 7), scores (T, M[, 1]), labels (T, M[, 1]) of ONE frame (label 0 = padding) -> fused
    (K,7) float64, (K,) float64, (K,) int, sorted by fused score (wbf_3d.py:118-203, iou_type '3d').  With obj_ids
    (T, M[, 1]): weighted_tracking_boxes_fusion_3d (wbf_3d.py:205-265), a fourth result (K,) int of object ids."""
    boxes = np.asarray(boxes)
    ids_in = None if obj_ids is None else np.asarray(obj_ids).reshape(boxes.shape[0], -1)
    scores = np.asarray(scores).reshape(boxes.shape[0], -1)
    labels = np.asarray(labels).reshape(boxes.shape[0], -1)
    t = boxes.shape[0]
    weights = np.ones(t) if weights is None else np.array(weights)
    per_label = {}
    for i in range(t):                                                          # prefilter_boxes :10-51
        for j in range(boxes.shape[1]):
            lab = int(labels[i][j])
            if lab == 0:
                continue
            row = [lab, float(scores[i][j]) * weights[i]] + ([] if ids_in is None else [int(ids_in[i][j])]) + [float(v) for v in boxes[i][j][:7]]
            per_label.setdefault(lab, []).append(row)
    for lab in per_label:
        arr = np.array(per_label[lab])
        arr = arr[arr[:, 1].argsort()[::-1]]
        per_label[lab] = arr[arr[:, 1] >= skip_box_thr[lab - 1]]
    empty = (np.zeros((0, 7)), np.zeros((0,)), np.zeros((0,))) + (() if ids_in is None else (np.zeros((0,)),))
    if len(per_label) == 0:
        return empty
    overall = []
    for lab, cand in per_label.items():
        groups, fused = [], []
        for j in range(len(cand)):
            idx = -1
            if fused:                                                           # find_matching_box :98-116
                ious = iou3d_one_to_many(cand[j][-7:], np.array(fused)[:, -7:])
                best = int(ious.argmax())
                if float(ious[best]) > iou_thr[lab - 1]:
                    idx = best
            if idx != -1:
                groups[idx].append(cand[j])
                fused[idx] = _fused(groups[idx], conf_type)
            else:
                groups.append([cand[j].copy()])
                fused.append(cand[j].copy())
        for i in range(len(groups)):                                            # :186-190
            n = len(groups[i])
            if not allows_overflow:
                fused[i][1] = fused[i][1] * min(weights.sum(), n) / weights.sum()
            else:
                fused[i][1] = fused[i][1] * n / weights.sum()
        if fused:
            overall.append(np.array(fused))
    if not overall:
        return empty
    overall = np.concatenate(overall, axis=0)
    overall = overall[overall[:, 1].argsort()[::-1]]
    if ids_in is None:
        return overall[:, -7:], overall[:, 1], overall[:, 0].astype(int)
    return overall[:, 3:], overall[:, 1], overall[:, 0].astype(int), overall[:, 2].astype(int)
