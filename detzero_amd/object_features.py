"""Cropped object points -> GRM / PRM model inputs on the GPU (second half).

Inference-time mirror of the feature extraction the reference's refining datasets run per object in numpy
(``WaymoGeometryDataset.extract_track_feature`` refining/detzero_refine/datasets/waymo/waymo_geometry_dataset.py:26-155,
``WaymoPositionDataset.extract_track_feature`` waymo_position_dataset.py:31-184) followed by ``collate_batch``
(datasets/dataset.py:207-258) and ``.float().cuda()``: here a whole batch of object tracks is packed once, the fixed-size
point selection is drawn on the host exactly like ``sample_points`` does (utils/data_utils.py:12-30 - one
``random.sample`` call per over-full set, in the reference's order, so a run seeded like the reference keeps the same
points), and one pass of ``dz_grm_encode_points`` / ``dz_prm_encode_points`` writes the float32 model inputs.

A track is what the crop step stores per object (daemon/prepare_object_data.py:274-313, object_crop.py):
``{'boxes_global' (T,7) float64, 'score' (T,), 'pts': list of T (n_i,4) float64 [x,y,z global, tanh(intensity)], 'name'}``.
The result dicts feed ``refine_modules.GeometryTransformer`` / ``PositionTransformer`` directly.
"""
import random

import numpy as np
import torch

from . import lib as L

CLASS_ID = {'Vehicle': 1, 'Pedestrian': 2, 'Cyclist': 3}
GRM_FLAGS = {'xyz': 1, 'intensity': 2, 'p2s': 4, 'score': 8}
PRM_CODES = {'xyz': 0, 'intensity': 1, 'p2co': 2, 'score': 3, 'class': 4}


class PackedTracks:
    """Device copy of a batch of tracks: points of all boxes back to back (object-major, frame order)."""

    def __init__(self, tracks, device=None):
        dev = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.device = dev
        self.batch = len(tracks)
        if self.batch == 0:
            raise L.DetZeroHipError('PackedTracks: empty batch of object tracks')
        self.counts = [[int(p.shape[0]) for p in t['pts']] for t in tracks]           # points per box, per object
        boxes = [np.asarray(t['boxes_global'], dtype=np.float64)[:, :7] for t in tracks]
        for t, b, c in zip(tracks, boxes, self.counts):
            if not (b.shape[0] == len(c) == len(t['score'])):
                raise L.DetZeroHipError('object track with %d boxes, %d scores, %d point sets' % (b.shape[0], len(t['score']), len(c)))
        self.box_num = [len(c) for c in self.counts]
        flat = [np.asarray(p, dtype=np.float64)[:, :4] for t in tracks for p in t['pts']]
        pts = np.concatenate(flat, axis=0) if flat else np.zeros((0, 4))
        if pts.shape[0] >= 2 ** 31:
            raise L.DetZeroHipError('more than 2^31 object points in one batch')
        box_off = np.zeros(len(flat) + 1, dtype=np.int32)
        np.cumsum([p.shape[0] for p in flat], out=box_off[1:])
        obj_off = np.zeros(self.batch + 1, dtype=np.int32)
        np.cumsum(self.box_num, out=obj_off[1:])
        self.obj_box_start = obj_off[:-1].tolist()
        self.scores = [np.asarray(t['score']) for t in tracks]
        self.classes = [CLASS_ID[t['name']] if isinstance(t.get('name'), str) else int(t.get('name', 0)) for t in tracks]

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.pts = up(pts if pts.shape[0] else np.zeros((1, 4)))
        self.box_offsets = up(box_off)
        self.traj_host = np.concatenate(boxes, axis=0) if boxes else np.zeros((1, 7))
        self.traj = up(self.traj_host)
        self.score = up(np.concatenate(self.scores).astype(np.float64) if boxes else np.zeros(1))
        self.obj_box_offsets = up(obj_off)
        self.obj_cls = up(np.asarray(self.classes, dtype=np.int32))


class DeviceDraw:
    """Pass as `rng` to grm_features / prm_features / crm_features to draw the fixed-size selections on the device
    (dz_draw_subsets): the distribution of sample_points, a counter-based stream keyed by (seed, set) instead of Python's
    random - no host loop over the boxes, nothing uploaded.  Streams: 1 = GRM memory sets (one per object), 2 = GRM query
    sets (object x query), 3 = PRM / CRM query sets (one per box), 4 = PRM memory sets (one per box)."""

    def __init__(self, seed=0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF

    def stream_seed(self, stream):
        return (self.seed ^ (0xA24BAED4963EE407 * stream)) & 0xFFFFFFFFFFFFFFFF

    def draw(self, counts, k, stream, device):
        """counts: per-set row counts (sequence or int32 tensor) -> (n_sets, k) int32 device tensor of kept rows / -1."""
        c = counts if torch.is_tensor(counts) else torch.as_tensor(np.asarray(counts, dtype=np.int32))
        c = c.to(device=device, dtype=torch.int32).contiguous()
        out = torch.empty((max(c.numel(), 1), k), dtype=torch.int32, device=device)
        with torch.cuda.device(device):
            rc = L.load().dz_draw_subsets(L.ptr(c), c.numel(), k, self.stream_seed(stream), 0, L.ptr(out), L.stream())
        L.check(rc, 'dz_draw_subsets')
        return out


def _draw(n, k, out, rng):
    """Row indices sample_points keeps (data_utils.py:12-30, replace=False) into out[:k]; the rest stays -1 (zero rows)."""
    if n >= k:
        idx = rng.sample(range(0, n), k)
        idx.sort()
        out[:k] = idx
    else:
        out[:n] = np.arange(n)


def grm_selection(packed, query_num=3, query_pts_num=256, memory_pts_num=4096, rng=random):
    """Host side of waymo_geometry_dataset.py:70-71,127-130 for every object, in the reference's drawing order
    (memory points first, then the queries by descending score).  Returns (mem_idx (B,mem_n), query_box (B,q_max),
    query_idx (B,q_max,q_n), geo_query_num list); q_max = the batch's largest query count (collate_batch pads to it)."""
    b = packed.batch
    orders = [np.argsort(s)[::-1][:query_num] for s in packed.scores]
    q_max = max([len(o) for o in orders], default=0)
    mem_idx = np.full((b, memory_pts_num), -1, dtype=np.int32)
    query_box = np.full((b, max(q_max, 1)), -1, dtype=np.int32)
    query_idx = np.full((b, max(q_max, 1), query_pts_num), -1, dtype=np.int32)
    for i in range(b):
        _draw(sum(packed.counts[i]), memory_pts_num, mem_idx[i], rng)
        for q, f in enumerate(orders[i]):
            query_box[i, q] = packed.obj_box_start[i] + int(f)
            _draw(packed.counts[i][int(f)], query_pts_num, query_idx[i, q], rng)
    return mem_idx, query_box, query_idx, [len(o) for o in orders], orders


def grm_features(tracks, encoding=('xyz', 'intensity', 'p2s', 'score'), query_num=3, query_pts_num=256, memory_pts_num=4096,
                 rng=random, device=None):
    """Batch of tracks -> {'geo_memory_points' (B,mem_n,C), 'geo_query_points' (B,q_max,q_n,4), 'geo_query_boxes' (B,q_max,7)
    float32 device tensors, 'geo_query_num', 'batch_size'} - the collated, device-resident GRM input."""
    if any(e not in GRM_FLAGS for e in encoding):
        raise L.DetZeroHipError('GRM encoding %r (supported: %s)' % (list(encoding), sorted(GRM_FLAGS)))
    packed = tracks if isinstance(tracks, PackedTracks) else PackedTracks(tracks, device)
    dev = packed.device
    flags = sum(GRM_FLAGS[e] for e in set(encoding))
    if isinstance(rng, DeviceDraw):
        orders = [np.argsort(sc)[::-1][:query_num] for sc in packed.scores]
        qnum = [len(o) for o in orders]
        q_max = max(max(qnum, default=0), 1)
        query_box = np.full((packed.batch, q_max), -1, dtype=np.int32)
        qcounts = np.zeros((packed.batch, q_max), dtype=np.int32)
        for i, o in enumerate(orders):
            for q, f in enumerate(o):
                query_box[i, q] = packed.obj_box_start[i] + int(f)
                qcounts[i, q] = packed.counts[i][int(f)]
        d_mem = rng.draw([sum(c) for c in packed.counts], memory_pts_num, 1, dev)
        d_qi = rng.draw(qcounts.reshape(-1), query_pts_num, 2, dev)
        d_qb = torch.from_numpy(query_box).to(dev)
    else:
        mem_idx, query_box, query_idx, qnum, orders = grm_selection(packed, query_num, query_pts_num, memory_pts_num, rng)
        d_mem, d_qb, d_qi = (torch.from_numpy(a).to(dev) for a in (mem_idx, query_box, query_idx))
    b, q_max = packed.batch, query_box.shape[1]
    lib = L.load()
    cm = lib.dz_grm_feature_channels(flags)
    memory = torch.empty((b, memory_pts_num, cm), dtype=torch.float32, device=dev)
    query = torch.empty((b, q_max, query_pts_num, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.dz_grm_encode_points(L.ptr(packed.pts), L.ptr(packed.box_offsets), L.ptr(packed.traj), L.ptr(packed.score),
                                      L.ptr(packed.obj_box_offsets), L.ptr(d_mem), memory_pts_num, L.ptr(d_qb), L.ptr(d_qi), q_max,
                                      query_pts_num, b, flags, L.ptr(memory), L.ptr(query), L.stream())
    L.check(rc, 'dz_grm_encode_points')
    # query boxes: centre and heading are zero in the box's own frame (waymo_geometry_dataset.py:82), sizes stay
    qboxes = np.zeros((b, q_max, 7), dtype=np.float32)
    for i in range(b):
        for q, f in enumerate(orders[i]):
            qboxes[i, q, 3:6] = packed.traj_host[packed.obj_box_start[i] + int(f), 3:6]
    return {'geo_memory_points': memory, 'geo_query_points': query, 'geo_query_boxes': torch.from_numpy(qboxes).to(dev),
            'geo_query_num': qnum, 'batch_size': b}


def prm_selection(packed, query_pts_num=256, memory_pts_num=48, rng=random):
    """Host side of waymo_position_dataset.py:85-90: per box a query draw, then a memory draw.  (F,q_n), (F,m_n) int32."""
    f = sum(packed.box_num)
    q_idx = np.full((max(f, 1), query_pts_num), -1, dtype=np.int32)
    m_idx = np.full((max(f, 1), memory_pts_num), -1, dtype=np.int32)
    k = 0
    for counts in packed.counts:
        for n in counts:
            _draw(n, query_pts_num, q_idx[k], rng)
            _draw(n, memory_pts_num, m_idx[k], rng)
            k += 1
    return q_idx, m_idx


def prm_features(tracks, encoding=('xyz', 'intensity', 'p2co', 'score'), query_num=200, query_pts_num=256, memory_pts_num=48,
                 rng=random, device=None):
    """Batch of tracks -> {'pos_query_points' (B,query_num,q_n,C), 'pos_memory_points' (B,query_num,m_n,C), 'pos_trajectory'
    (B,query_num,7), 'padding_mask' (B,query_num) float32, 'pos_init_box' (B,7) float64, 'box_num', 'obj_cls', 'batch_size'}."""
    if any(e not in PRM_CODES for e in encoding):
        raise L.DetZeroHipError('PRM encoding %r (supported: %s)' % (list(encoding), sorted(PRM_CODES)))
    packed = tracks if isinstance(tracks, PackedTracks) else PackedTracks(tracks, device)
    if max(packed.box_num, default=0) > query_num:
        raise L.DetZeroHipError('object track with %d boxes exceeds QUERY_NUM = %d' % (max(packed.box_num), query_num))
    dev = packed.device
    b = packed.batch
    if isinstance(rng, DeviceDraw):
        flat = [n for counts in packed.counts for n in counts]
        d_q, d_m = rng.draw(flat, query_pts_num, 3, dev), rng.draw(flat, memory_pts_num, 4, dev)
    else:
        q_idx, m_idx = prm_selection(packed, query_pts_num, memory_pts_num, rng)
        d_q, d_m = torch.from_numpy(q_idx).to(dev), torch.from_numpy(m_idx).to(dev)
    codes = np.asarray([PRM_CODES[e] for e in encoding], dtype=np.int32)
    lib = L.load()
    ch = lib.dz_prm_feature_channels(codes.ctypes.data, len(codes))
    query = torch.empty((b, query_num, query_pts_num, ch), dtype=torch.float32, device=dev)
    memory = torch.empty((b, query_num, memory_pts_num, ch), dtype=torch.float32, device=dev)
    traj_local = torch.empty((b, query_num, 7), dtype=torch.float32, device=dev)
    mask = torch.empty((b, query_num), dtype=torch.float32, device=dev)
    init_box = torch.empty((b, 7), dtype=torch.float64, device=dev)
    scratch = torch.empty((b * query_num * 27 + 2 * b,), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = lib.dz_prm_encode_points(L.ptr(packed.pts), L.ptr(packed.box_offsets), L.ptr(packed.traj), L.ptr(packed.score),
                                      L.ptr(packed.obj_box_offsets), L.ptr(packed.obj_cls), L.ptr(d_q), L.ptr(d_m), query_pts_num,
                                      memory_pts_num, b, query_num, codes.ctypes.data, len(codes), L.ptr(query), L.ptr(memory),
                                      L.ptr(traj_local), L.ptr(mask), L.ptr(init_box), L.ptr(scratch), L.stream())
    L.check(rc, 'dz_prm_encode_points')
    return {'pos_query_points': query, 'pos_memory_points': memory, 'pos_trajectory': traj_local, 'padding_mask': mask,
            'pos_init_box': init_box, 'box_num': list(packed.box_num), 'obj_cls': packed.classes, 'batch_size': b}


def crm_selection(packed, query_pts_num=256, rng=random):
    """Host side of waymo_confidence_dataset.py:106-111: one draw per box.  (F, q_n) int32."""
    q_idx = np.full((max(sum(packed.box_num), 1), query_pts_num), -1, dtype=np.int32)
    k = 0
    for counts in packed.counts:
        for n in counts:
            _draw(n, query_pts_num, q_idx[k], rng)
            k += 1
    return q_idx


def crm_features(tracks, encoding=('xyz', 'intensity', 'p2co', 'score'), query_num=200, query_pts_num=256, rng=random, device=None):
    """Batch of tracks -> {'conf_points' (B,query_num,q_n,C) float32 device tensor, 'conf_score' (B,query_num) float64 numpy
    (padded with -1), 'box_num', 'batch_size'} - the collated CRM input (waymo_confidence_dataset.py:59-162; the per-point
    encoding is the PRM query encoding, same kernel)."""
    if any(e not in PRM_CODES or e == 'class' for e in encoding):
        raise L.DetZeroHipError('CRM encoding %r (supported on the device: xyz, intensity, p2co, score)' % (list(encoding),))
    packed = tracks if isinstance(tracks, PackedTracks) else PackedTracks(tracks, device)
    if max(packed.box_num, default=0) > query_num:
        raise L.DetZeroHipError('object track with %d boxes exceeds QUERY_NUM = %d' % (max(packed.box_num), query_num))
    dev, b = packed.device, packed.batch
    if isinstance(rng, DeviceDraw):
        d_q = rng.draw([n for counts in packed.counts for n in counts], query_pts_num, 3, dev)
    else:
        d_q = torch.from_numpy(crm_selection(packed, query_pts_num, rng)).to(dev)
    codes = np.asarray([PRM_CODES[e] for e in encoding], dtype=np.int32)
    lib = L.load()
    ch = lib.dz_prm_feature_channels(codes.ctypes.data, len(codes))
    points = torch.empty((b, query_num, query_pts_num, ch), dtype=torch.float32, device=dev)
    traj_local = torch.empty((b, query_num, 7), dtype=torch.float32, device=dev)
    mask = torch.empty((b, query_num), dtype=torch.float32, device=dev)
    init_box = torch.empty((b, 7), dtype=torch.float64, device=dev)
    scratch = torch.empty((b * query_num * 27 + 2 * b,), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = lib.dz_prm_encode_points(L.ptr(packed.pts), L.ptr(packed.box_offsets), L.ptr(packed.traj), L.ptr(packed.score),
                                      L.ptr(packed.obj_box_offsets), None, L.ptr(d_q), None, query_pts_num, 0, b, query_num,
                                      codes.ctypes.data, len(codes), L.ptr(points), None, L.ptr(traj_local), L.ptr(mask),
                                      L.ptr(init_box), L.ptr(scratch), L.stream())
    L.check(rc, 'dz_prm_encode_points')
    score = np.full((b, query_num), -1.0)
    for i, sc in enumerate(packed.scores):
        score[i, :len(sc)] = sc
    return {'conf_points': points, 'conf_score': score, 'box_num': list(packed.box_num), 'batch_size': b}
