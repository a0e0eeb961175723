"""ORACLE (test infrastructure only): CPU restatement of the per-object feature encoding that turns cropped object
points into the inputs of the refining models, INFERENCE branch only (training-time sampling / augmentation is out of
scope):

  GRM  refining/detzero_refine/datasets/waymo/waymo_geometry_dataset.py:26-155 (extract_track_feature)
  PRM  refining/detzero_refine/datasets/waymo/waymo_position_dataset.py:31-184 (extract_track_feature)
  CRM  refining/detzero_refine/datasets/waymo/waymo_confidence_dataset.py:59-162 (extract_track_feature)
  helpers  refining/detzero_refine/utils/data_utils.py:6-10 (rotate_yaw), :12-30 (sample_points), :33-42
           (limit_heading_range), :62-71 (local_coords_transform), :74-113 (init_coords_transform);
           utils/detzero_utils/box_utils.py:28-53 (boxes_to_corners_3d, float32 through torch),
           common_utils.py:220-244 (rotate_points_along_z)
  batch    refining/detzero_refine/datasets/dataset.py:207-258 (collate_batch: zero padding of the GRM queries)

Pinned by tests/golden/refine_feat_golden.npz (GRM, PRM) and crm_golden.npz (CRM), which tests/golden/gen_refine_feat_golden.py /
gen_crm_golden.py produced by running the
reference's own dataset classes on seeded synthetic tracks (same Python `random` stream: sample_points draws with
random.sample, so the oracle draws the same subsets when seeded identically).

Arithmetic as in the reference: points and boxes are float64, the yaw matrix is float32 (rotate_yaw builds it with
dtype=np.float32), box corners are float32 end to end (torch), features are float64 and only become float32 at the
model input (`.float()`).
"""
import random

import numpy as np

CLASS_ID = {'Vehicle': 1, 'Pedestrian': 2, 'Cyclist': 3}


def yaw_matrix(yaw):
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]], dtype=np.float32)        # data_utils.py:6-10


def draw_subset(n, k, rng=random):
    """Index list of sample_points (data_utils.py:12-30, replace=False): a sorted random k-subset when n >= k (one
    random.sample call), else all n rows; the caller zero-pads to k."""
    if n >= k:
        idx = rng.sample(range(0, n), k)
        idx.sort()
        return np.asarray(idx, dtype=np.int64)
    return np.arange(n, dtype=np.int64)


def take_padded(rows, idx, k):
    out = np.zeros((k, rows.shape[1]), dtype=np.float64)
    out[:len(idx)] = rows[idx]
    return out


def wrap_heading(a):
    """limit_heading_range (data_utils.py:33-42): repeated +-2*pi steps into [-pi, pi)."""
    a = np.array(a, dtype=np.float64, copy=True)
    while (a >= np.pi).any():
        a[a >= np.pi] -= 2 * np.pi
    while (a < -np.pi).any():
        a[a < -np.pi] += 2 * np.pi
    return a


def corners_f32(boxes):
    """box_utils.py:28-53 on a numpy input: everything in float32."""
    b = np.asarray(boxes, dtype=np.float64).astype(np.float32)
    tmpl = np.array([[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1],
                     [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]], dtype=np.float32) / np.float32(2)
    c = b[:, None, 3:6] * tmpl[None]
    ca, sa = np.cos(b[:, 6]).astype(np.float32), np.sin(b[:, 6]).astype(np.float32)
    rot = np.zeros((b.shape[0], 3, 3), dtype=np.float32)
    rot[:, 0, 0], rot[:, 0, 1], rot[:, 1, 0], rot[:, 1, 1], rot[:, 2, 2] = ca, sa, -sa, ca, 1
    c = np.einsum('nij,njk->nik', c, rot).astype(np.float32)
    return c + b[:, None, 0:3]


# ------------------------------------------------------------------------------------------------ GRM
def grm_object(track, encoding=('xyz', 'intensity', 'p2s', 'score'), query_num=3, query_pts_num=256, memory_pts_num=4096,
               rng=random):
    """track: {'boxes_global' (T,7), 'score' (T,), 'pts' list of T (n_i,4) float64}.  Returns the inference-time
    geo_* entries of one object (waymo_geometry_dataset.py:58-139)."""
    traj = np.asarray(track['boxes_global'], dtype=np.float64)
    score = np.asarray(track['score'])
    order = np.argsort(score)[::-1][:query_num]                                                 # :70-71
    local = []
    for i, p in enumerate(track['pts']):                                                        # :75 -> data_utils.py:62-71
        q = np.array(p, dtype=np.float64, copy=True)
        q[:, :3] = (q[:, :3] - traj[i, :3]) @ yaw_matrix(traj[i, 6]).T
        local.append(q)
    query_box = traj[order].copy()
    query_box[:, [0, 1, 2, 6]] = 0                                                              # :82
    feats = []
    for i, q in enumerate(local):                                                               # :91-116
        cols = []
        if 'xyz' in encoding:
            cols.append(q[:, :3])
        if 'intensity' in encoding:
            cols.append(q[:, [3]])
        if 'p2s' in encoding:
            cols.append(traj[i, 3:6] / 2 - q[:, :3])
            cols.append(traj[i, 3:6] / 2 + q[:, :3])
        if 'score' in encoding:
            cols.append(np.full((q.shape[0], 1), score[i], dtype=np.float64))
        feats.append(np.concatenate(cols, axis=1))
    feats = np.concatenate(feats, axis=0)
    memory = take_padded(feats, draw_subset(feats.shape[0], memory_pts_num, rng), memory_pts_num)   # :127
    queries = []
    for ind in order:                                                                               # :129-130
        q = local[ind]
        queries.append(take_padded(q, draw_subset(q.shape[0], query_pts_num, rng), query_pts_num))
    return {'geo_query_num': len(order), 'geo_query_boxes': query_box, 'geo_query_points': np.array(queries),
            'geo_memory_points': memory, 'geo_trajectory': traj, 'geo_score': score}


def grm_batch(objs):
    """dataset.py:207-258: stack, zero-padding queries to the batch's largest geo_query_num."""
    qmax = max(o['geo_query_num'] for o in objs)
    qp, qb = [], []
    for o in objs:
        p, b = o['geo_query_points'], o['geo_query_boxes']
        qp.append(np.concatenate([p, np.zeros((qmax - p.shape[0],) + p.shape[1:])], axis=0))
        qb.append(np.concatenate([b, np.zeros((qmax - b.shape[0], b.shape[1]))], axis=0))
    return {'geo_query_num': [o['geo_query_num'] for o in objs], 'geo_query_points': np.stack(qp), 'geo_query_boxes': np.stack(qb),
            'geo_memory_points': np.stack([o['geo_memory_points'] for o in objs]), 'batch_size': len(objs)}


# ------------------------------------------------------------------------------------------------ PRM
def prm_object(track, encoding=('xyz', 'intensity', 'p2co', 'score'), query_num=200, query_pts_num=256, memory_pts_num=48,
               rng=random):
    """track as in grm_object plus 'name'.  Returns the inference-time pos_* entries (waymo_position_dataset.py:66-178)."""
    traj = np.array(track['boxes_global'], dtype=np.float64, copy=True)[:, :7]
    score = np.asarray(track['score'])
    t = traj.shape[0]
    init = traj[t // 2].copy()                                                                  # :72-73
    init[6] = wrap_heading(init[[6]])[0]                                                        # data_utils.py:80
    rot = yaw_matrix(init[6]).T
    pts = []
    for p in track['pts']:                                                                      # data_utils.py:83-85
        q = np.array(p, dtype=np.float64, copy=True)
        q[:, :3] = (q[:, :3] - init[:3]) @ rot
        pts.append(q)
    traj[:, 6] = wrap_heading(traj[:, 6])                                                       # data_utils.py:88-96
    traj[:, :3] = (traj[:, :3] - init[:3]) @ rot
    traj[:, 6] -= init[6]
    traj[:, 6] = wrap_heading(traj[:, 6])
    qs, ms = [], []
    for q in pts:                                                                               # :85-90 (query draw first, then memory)
        qs.append(take_padded(q, draw_subset(q.shape[0], query_pts_num, rng), query_pts_num))
        ms.append(take_padded(q, draw_subset(q.shape[0], memory_pts_num, rng), memory_pts_num))
    qs, ms = np.stack(qs), np.stack(ms)
    cls = CLASS_ID[track['name']] if isinstance(track['name'], str) else int(track['name'])
    lq, lm = [], []
    for e in encoding:                                                                          # :97-136
        if e == 'xyz':
            lq.append(qs[:, :, :3]); lm.append(ms[:, :, :3])
        elif e == 'intensity':
            lq.append(qs[:, :, [3]]); lm.append(ms[:, :, [3]])
        elif e == 'p2co':
            anchor = np.concatenate([corners_f32(traj).reshape(t, -1), traj[:, :3]], axis=-1)     # (T, 27)
            lq.append(np.tile(qs[:, :, :3], (1, 1, 9)) - anchor[:, None, :])
            lm.append(np.tile(ms[:, :, :3], (1, 1, 9)) - anchor[:, None, :])
        elif e == 'score':
            lq.append(np.tile(score[:, None, None], (1, query_pts_num, 1)))
            lm.append(np.tile(score[:, None, None], (1, memory_pts_num, 1)))
        elif e == 'class':
            one = np.zeros(3)
            one[cls - 1] = 1
            lq.append(np.tile(one[None, None, :], (t, query_pts_num, 1)))
            lm.append(np.tile(one[None, None, :], (t, memory_pts_num, 1)))
        else:
            raise NotImplementedError(e)
    lq, lm = np.concatenate(lq, axis=2), np.concatenate(lm, axis=2)
    lq = np.concatenate([lq, np.zeros((query_num - t, query_pts_num, lq.shape[2]))], axis=0)      # :146-155
    lm = np.concatenate([lm, np.zeros((query_num - t, memory_pts_num, lm.shape[2]))], axis=0)
    traj_pad = np.concatenate([traj, np.zeros((query_num - t, 7), dtype=np.float32)], axis=0)
    mask = np.concatenate([np.zeros(t), np.ones(query_num - t)])
    return {'pos_trajectory': traj_pad, 'pos_scores': score, 'pos_init_box': init, 'box_num': t, 'padding_mask': mask,
            'pos_query_points': lq, 'pos_memory_points': lm, 'obj_cls': cls}


def prm_batch(objs):
    keys = ['pos_trajectory', 'pos_init_box', 'padding_mask', 'pos_query_points', 'pos_memory_points', 'obj_cls']
    out = {k: np.stack([o[k] for o in objs]) for k in keys}
    out['box_num'] = [o['box_num'] for o in objs]
    out['batch_size'] = len(objs)
    return out


# ------------------------------------------------------------------------------------------------ CRM
def crm_object(track, encoding=('xyz', 'intensity', 'p2co', 'score'), query_num=200, query_pts_num=256, rng=random):
    """Inference-time conf_* entries of one object (waymo_confidence_dataset.py:82-160): the PRM query-point encoding in
    the frame of the middle box (one draw per box), plus 'box_pos'; scores padded with -1."""
    traj = np.array(track['boxes_global'], dtype=np.float64, copy=True)[:, :7]
    score = np.asarray(track['score'])
    t = traj.shape[0]
    init = traj[t // 2].copy()
    init[6] = wrap_heading(init[[6]])[0]
    rot = yaw_matrix(init[6]).T
    pts = []
    for p in track['pts']:
        q = np.array(p, dtype=np.float64, copy=True)
        q[:, :3] = (q[:, :3] - init[:3]) @ rot
        pts.append(q)
    traj[:, 6] = wrap_heading(traj[:, 6])
    traj[:, :3] = (traj[:, :3] - init[:3]) @ rot
    traj[:, 6] -= init[6]
    traj[:, 6] = wrap_heading(traj[:, 6])
    qs = np.stack([take_padded(q, draw_subset(q.shape[0], query_pts_num, rng), query_pts_num) for q in pts])
    cols = []
    for e in encoding:
        if e == 'xyz':
            cols.append(qs[:, :, :3])
        elif e == 'intensity':
            cols.append(qs[:, :, [3]])
        elif e == 'p2co':
            anchor = np.concatenate([corners_f32(traj).reshape(t, -1), traj[:, :3]], axis=-1)
            cols.append(np.tile(qs[:, :, :3], (1, 1, 9)) - anchor[:, None, :])
        elif e == 'box_pos':
            cols.append(np.tile(np.concatenate([traj[:, :3], traj[:, 6:7]], axis=-1)[:, None, :], (1, query_pts_num, 1)))
        elif e == 'score':
            cols.append(np.tile(score[:, None, None], (1, query_pts_num, 1)))
        else:
            raise NotImplementedError(e)
    feat = np.concatenate(cols, axis=2)
    feat = np.concatenate([feat, np.zeros((query_num - t, query_pts_num, feat.shape[2]))], axis=0)
    return {'conf_points': feat, 'conf_score': np.concatenate((score, np.full(query_num - t, -1))), 'box_num': t}


def crm_batch(objs):
    return {'conf_points': np.stack([o['conf_points'] for o in objs]), 'conf_score': np.stack([o['conf_score'] for o in objs]),
            'box_num': [o['box_num'] for o in objs], 'batch_size': len(objs)}


# ------------------------------------------------------------------------------------------------ device-side draw (restated)
_M64 = (1 << 64) - 1


def device_draw_hash(seed, set_id, i):
    """csrc/object_features.hip draw_hash (splitmix64 finaliser of seed + C1 (set+1) + C2 i), upper 32 bits."""
    z = (seed + 0x9E3779B97F4A7C15 * (set_id + 1) + i * 0xD1B54A32D192ED03) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    z ^= z >> 31
    return z >> 32


def device_draw_subset(n, k, seed, set_id):
    """dz_draw_subsets for one set: selection sampling, row i kept when floor(r (n - i) / 2^32) < k - kept."""
    if n < k:
        return np.arange(n, dtype=np.int64)
    out = []
    for i in range(n):
        if ((device_draw_hash(seed, set_id, i) * (n - i)) >> 32) < k - len(out):
            out.append(i)
            if len(out) == k:
                break
    return np.asarray(out, dtype=np.int64)


def device_stream_seed(seed, stream):
    return ((seed & _M64) ^ (0xA24BAED4963EE407 * stream)) & _M64


class DeviceDrawReplay:
    """Stand-in for `random` that replays the device draws in the order the oracle's *_object functions ask for them:
    queue of (stream seed, set id) prepared by the caller, one per rng.sample call."""

    def __init__(self, queue):
        self.queue = list(queue)

    def sample(self, population, k):
        seed, set_id = self.queue.pop(0)
        return [int(v) for v in device_draw_subset(len(population), k, seed, set_id)]
