"""Test-time augmentation on the GPU.

Mirror of the reference's TTA path with every step on the device:

* ``TestTimeAugmentor`` - the config-driven list of augmented copies of a frame
  (detection/detzero_det/datasets/augmentor/test_time_augmentor.py:10-100; TEST_TIME_AUGMENTOR block of
  det_dataset_cfgs/waymo_1sweep.yaml:48-58); the copies' names are the reference's ``tta_*`` keys;
* ``restore_boxes`` - boxes of the copies back to the original frame (models/centerpoint.py:165-203);
* ``wbf_online`` - weighted box fusion with the reference's online thresholds (utils/ensemble_utils/ensemble.py:7-33 ->
  wbf_3d.py:118-203), one device pass instead of one IoU launch + sync per candidate box;
* ``TTAPipeline`` - a ``FramePipeline`` run over the copies of each frame, followed by the two steps above: the
  reference's ``CenterPoint.test_time_augment`` for a batch of frames.
"""
import numpy as np
import torch

from . import lib as L
from .lib import DetZeroHipError

KINDS = {'original': 0, 'flip_x': 1, 'flip_y': 2, 'flip_xy': 3, 'rot': 4, 'scale': 5}
IOU_THR = (0.8, 0.6, 0.7)              # ensemble.py:15-16
SKIP_BOX_THR = (0.1, 0.01, 0.01)


def parse_op(name):
    """'tta_original' | 'tta_flip_x' | 'tta_rot_-0.78539816' | 'tta_scale_0.95' -> (kind code, float parameter)."""
    parts = name.split('_')
    if len(parts) == 2 and parts[1] == 'original':
        return 0, 0.0
    if len(parts) != 3 or parts[0] != 'tta':
        raise DetZeroHipError('unknown TTA operation %r' % name)
    if parts[1] == 'flip' and parts[2] in ('x', 'y', 'xy'):
        return KINDS['flip_' + parts[2]], 0.0
    if parts[1] in ('rot', 'scale'):
        return KINDS[parts[1]], float(parts[2])
    raise DetZeroHipError('unknown TTA operation %r' % name)


def _op_arrays(names):
    kinds, params = zip(*[parse_op(n) for n in names])
    return np.asarray(kinds, dtype=np.int32), np.asarray(params, dtype=np.float32)


class TestTimeAugmentor:
    """test_time_augmentor.py:10-100.  ``augmentor_configs``: the AUG_CONFIG_LIST (list of dicts / AttrDicts with NAME in
    world_flip / world_rotation / world_scaling) or an object with AUG_CONFIG_LIST and DISABLE_AUG_LIST."""
    __test__ = False

    def __init__(self, augmentor_configs, logger=None):
        cfgs = augmentor_configs if isinstance(augmentor_configs, (list, tuple)) else augmentor_configs.AUG_CONFIG_LIST
        disabled = [] if isinstance(augmentor_configs, (list, tuple)) else list(getattr(augmentor_configs, 'DISABLE_AUG_LIST', []))
        self.op_names = ['tta_original']
        for cfg in cfgs:
            name = cfg['NAME']
            if name in disabled:
                continue
            if name == 'world_flip':
                self.op_names += ['tta_flip_%s' % a for a in cfg['ALONG_AXIS_LIST']]
            elif name == 'world_rotation':
                self.op_names += ['tta_rot_%s' % str(a) for a in cfg['ROT_ANGLE'] if a != 0.]
            elif name == 'world_scaling':
                self.op_names += ['tta_scale_%s' % str(f) for f in cfg['SCALE_RANGE'] if f != 1.]
            else:
                raise DetZeroHipError('TestTimeAugmentor: unknown augmentor %r' % name)
        self.kinds, self.params = _op_arrays(self.op_names)

    def augment(self, points):
        """points (N,C) float32 device tensor -> (T,N,C): copy i = the frame under self.op_names[i]."""
        L.require_cuda(points)
        n, c = points.shape
        out = torch.empty((len(self.op_names), n, c), dtype=torch.float32, device=points.device)
        with torch.cuda.device(points.device):
            rc = L.load().dz_tta_augment_points(L.ptr(points), n, c, self.kinds.ctypes.data, self.params.ctypes.data, len(self.op_names),
                                                L.ptr(out), L.stream())
        L.check(rc, 'dz_tta_augment_points')
        return out

    def forward(self, data_dict):
        """The reference's dict-of-dicts form: {'tta_original': data_dict, 'tta_flip_x': {... 'points': ...}, ...}."""
        copies = self.augment(data_dict['points'])
        out = {}
        for i, name in enumerate(self.op_names):
            d = dict(data_dict)
            d['points'] = copies[i]
            out[name] = d
        return out


def restore_boxes(boxes, op_names):
    """boxes (F, T, M, >=7) float32 device tensor, copy axis in the order of op_names -> restored in place (and returned)."""
    L.require_cuda(boxes)
    f, t, m, dim = boxes.shape
    if t != len(op_names):
        raise DetZeroHipError('restore_boxes: %d copies, %d operations' % (t, len(op_names)))
    kinds, params = _op_arrays(op_names)
    with torch.cuda.device(boxes.device):
        rc = L.load().dz_tta_restore_boxes(L.ptr(boxes), f, t, m, dim, kinds.ctypes.data, params.ctypes.data, L.stream())
    L.check(rc, 'dz_tta_restore_boxes')
    return boxes


def wbf_fuse_nosync(boxes, scores, labels, n_models, weights=None, iou_thr=IOU_THR, skip_box_thr=SKIP_BOX_THR, conf_type='avg',
                    allows_overflow=False, obj_ids=None):
    """boxes (F, cand, 7) float32, scores (F, cand) float32, labels (F, cand) int32 (0 = padding), candidates model-major
    (cand = n_models x per-model rows).  -> (boxes (F,cand,7) float64, scores (F,cand) float64, labels (F,cand) int32,
    counts (F,) int32): the first counts[f] rows of frame f, sorted by fused score.  No host synchronisation.
    With obj_ids (F, cand) int32 the tracking variant (weighted_tracking_boxes_fusion_3d): a fifth result, the fused boxes'
    object ids (F, cand) int32, sits before the counts."""
    L.require_cuda(boxes, scores, labels, obj_ids)
    f, cand = scores.shape
    if cand % n_models:
        raise DetZeroHipError('wbf: %d candidates do not split into %d models' % (cand, n_models))
    if conf_type not in ('avg', 'max'):
        conf_type = 'avg'                                    # wbf_3d.py:148-150
    dev = boxes.device
    if weights is not None and len(weights) != n_models:
        # wbf_3d.py:143-145: a weights list of the wrong length is replaced by all ones (the reference prints a warning)
        import warnings
        warnings.warn('wbf: %d weights for %d models - using equal weights' % (len(weights), n_models))
        weights = None
    w = None if weights is None else torch.as_tensor(np.asarray(weights, dtype=np.float64)).to(dev)
    wsum = float(n_models) if weights is None else float(np.asarray(weights, dtype=np.float64).sum())
    lib = L.load()
    ws = torch.empty((lib.dz_wbf_workspace_bytes(f, cand) // 8 + 1,), dtype=torch.float64, device=dev)
    ob = torch.empty((f, cand, 7), dtype=torch.float64, device=dev)
    osc = torch.empty((f, cand), dtype=torch.float64, device=dev)
    ol = torch.empty((f, cand), dtype=torch.int32, device=dev)
    oc = torch.empty((f,), dtype=torch.int32, device=dev)
    oo = None if obj_ids is None else torch.empty((f, cand), dtype=torch.int32, device=dev)
    thr = np.asarray(iou_thr, dtype=np.float64)
    skip = np.asarray(skip_box_thr, dtype=np.float64)
    with torch.cuda.device(dev):
        rc = lib.dz_wbf_fuse_3d(L.ptr(boxes), L.ptr(scores), L.ptr(labels), L.ptr(obj_ids), f, cand, cand // n_models, L.ptr(w), n_models,
                                thr.ctypes.data, skip.ctypes.data, wsum, 1 if conf_type == 'max' else 0, 1 if allows_overflow else 0,
                                L.ptr(ob), L.ptr(osc), L.ptr(ol), L.ptr(oo), L.ptr(oc), L.ptr(ws), ws.numel() * 8, L.stream())
    L.check(rc, 'dz_wbf_fuse_3d')
    return (ob, osc, ol, oc) if obj_ids is None else (ob, osc, ol, oo, oc)


def wbf_online(boxes, scores, labels):
    """ensemble.py:7-33 for ONE frame: boxes (T, M, 7), scores (T, M[, 1]), labels (T, M[, 1]) device tensors ->
    (boxes (K,7) float64, scores (K,) float64, labels (K,) int64) on the device, sorted by fused score."""
    t, m = boxes.shape[0], boxes.shape[1]
    b = boxes[..., :7].float().reshape(1, t * m, 7).contiguous()
    s = scores.float().reshape(1, t * m).contiguous()
    la = labels.reshape(1, t * m).to(torch.int32).contiguous()
    ob, osc, ol, oc = wbf_fuse_nosync(b, s, la, t)
    k = int(oc.item())
    return ob[0, :k], osc[0, :k], ol[0, :k].long()


def wbf_tracking_v1(boxes, scores, labels, obj_ids):
    """ensemble.py:35-62 for ONE frame: as wbf_online, plus obj_ids (T, M[, 1]) -> (boxes, scores, labels, obj_ids (K,) int64)."""
    t, m = boxes.shape[0], boxes.shape[1]
    b = boxes[..., :7].float().reshape(1, t * m, 7).contiguous()
    s = scores.float().reshape(1, t * m).contiguous()
    la = labels.reshape(1, t * m).to(torch.int32).contiguous()
    ids = obj_ids.reshape(1, t * m).to(torch.int32).contiguous()
    ob, osc, ol, oo, oc = wbf_fuse_nosync(b, s, la, t, obj_ids=ids)
    k = int(oc.item())
    return ob[0, :k], osc[0, :k], ol[0, :k].long(), oo[0, :k].long()


class TTAPipeline:
    """CenterPoint.test_time_augment (centerpoint.py:131-208) for a batch of frames: every frame is run through the
    detector once per augmented copy (the copies are ordinary frames of the FramePipeline's batch), the copies' boxes are
    restored and fused.  ``__call__(frames)`` -> (boxes (F,cand,7) float64, scores (F,cand) float64, labels (F,cand) int32,
    counts (F,) int32), cand = copies x MAX_OBJ rows; sync-free."""

    def __init__(self, pipeline, augmentor):
        self.pipe = pipeline
        self.aug = augmentor

    @torch.no_grad()
    def __call__(self, frames):
        frames = [frames] if torch.is_tensor(frames) and frames.dim() == 2 else list(frames)
        t = len(self.aug.op_names)
        copies = [self.aug.augment(f) for f in frames]                      # F x (T,N,C)
        batch = [c[i] for c in copies for i in range(t)]                    # frame-major, copy-minor
        out, counts = self.pipe(batch)                                      # (F*T, K, 9), (F*T,)
        f, k = len(frames), out.shape[1]
        valid = torch.arange(k, device=out.device)[None, :] < counts[:, None].to(out.device)
        boxes = out[..., :7].reshape(f, t, k, 7).contiguous()
        restore_boxes(boxes, self.aug.op_names)
        scores = out[..., 7].reshape(f, t * k).contiguous()
        labels = torch.where(valid, out[..., 8].to(torch.int32), torch.zeros((), dtype=torch.int32, device=out.device))
        return wbf_fuse_nosync(boxes.reshape(f, t * k, 7), scores, labels.reshape(f, t * k).contiguous(), t)
