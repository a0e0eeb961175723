"""ORACLE (test infrastructure only): BEV backbone, CenterHead, decode and NMS on CPU torch.

Reference (paths under /root/reference/detection/detzero_det):
  * BaseBEVBackbone.forward  - models/centerpoint_modules/backbone2d.py:89-120 (layers :33-80)
  * CenterHead convs         - models/centerpoint_modules/center_head.py:14-48, 81-102, 440-447
  * decode                   - utils/centernet_utils.py:138-230 ; caller center_head.py:315-368
  * class-agnostic NMS       - utils/model_nms_utils.py:6-25 -> iou3d_nms_utils.nms_gpu
                               (utils/detzero_utils/ops/iou3d_nms/iou3d_nms_utils.py:154-170)
Pinned against the reference's own Python modules via tests/golden/*.npz (gen_golden.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import cref

HEAD_ORDER = ('center', 'center_z', 'dim', 'rot', 'iou', 'hm')
HEAD_CH = {'center': 2, 'center_z': 1, 'dim': 3, 'rot': 2, 'iou': 1, 'hm': 3}


def bn2d_eval(x, sd, prefix, eps):
    w, b = sd[prefix + '.weight'], sd[prefix + '.bias']
    m, v = sd[prefix + '.running_mean'], sd[prefix + '.running_var']
    return F.batch_norm(x, m, v, w, b, False, 0.0, eps)


def bev_backbone_forward(sd, x, layer_nums=(5, 5), layer_strides=(1, 2), upsample_strides=(1, 2),
                         prefix='backbone2d.'):
    ups = []
    for lvl in range(len(layer_nums)):
        p = '%sblocks.%d.' % (prefix, lvl)
        x = F.pad(x, (1, 1, 1, 1))
        x = F.conv2d(x, sd[p + '1.weight'], None, stride=layer_strides[lvl])
        x = torch.relu(bn2d_eval(x, sd, p + '2', 1e-3))
        for k in range(layer_nums[lvl]):
            x = F.conv2d(x, sd[p + '%d.weight' % (4 + 3 * k)], None, padding=1)
            x = torch.relu(bn2d_eval(x, sd, p + '%d' % (5 + 3 * k), 1e-3))
        d = '%sdeblocks.%d.' % (prefix, lvl)
        s = upsample_strides[lvl]
        u = F.conv_transpose2d(x, sd[d + '0.weight'], None, stride=s)
        ups.append(torch.relu(bn2d_eval(u, sd, d + '1', 1e-3)))
    return torch.cat(ups, dim=1)


def center_head_forward(sd, x, prefix='dense_head.', head_order=HEAD_ORDER):
    p = prefix + 'shared_conv.'
    x = F.conv2d(x, sd[p + '0.weight'], sd.get(p + '0.bias'), padding=1)
    x = torch.relu(bn2d_eval(x, sd, p + '1', 1e-5))
    out = {}
    for name in head_order:
        h = '%sheads_list.0.%s.' % (prefix, name)
        y = F.conv2d(x, sd[h + '0.0.weight'], sd.get(h + '0.0.bias'), padding=1)
        y = torch.relu(bn2d_eval(y, sd, h + '0.1', 1e-5))
        out[name] = F.conv2d(y, sd[h + '1.weight'], sd[h + '1.bias'], padding=1)
    return out


def topk_desc(scores_flat, k):
    """Deterministic top-k: score descending, ties by ascending flat index (torch.topk leaves ties
    implementation-defined; tests avoid exact ties, this only makes the oracle reproducible)."""
    s = scores_flat.numpy()
    order = np.lexsort((np.arange(s.size), -s.astype(np.float64)))
    order = order[:k]
    return torch.from_numpy(s[order].copy()), torch.from_numpy(order.astype(np.int64))


def decode(pred, pc_range, voxel_size, stride, k, score_thresh, post_center_limit_range, iou_weight=1):
    """center_head.py:325-344 + centernet_utils.py:138-230 for ONE head, batch entries separately.
    Returns a list (per batch item) of dicts with pred_boxes (n,7), pred_scores, pred_labels(0-based)."""
    hm = pred['hm'].sigmoid()
    dim = pred['dim'].exp()
    rot_cos, rot_sin = pred['rot'][:, 0:1], pred['rot'][:, 1:2]
    b, ncls, h, w = hm.shape
    scores = hm.flatten(2, 3)
    if iou_weight > 0:
        iou = torch.clamp(pred['iou'].reshape(b, 1, h * w), min=0, max=1)
        scores = scores * torch.pow(iou, 2)
    lim = torch.tensor(post_center_limit_range, dtype=torch.float32)
    res = []
    for bi in range(b):
        flat = scores[bi].reshape(-1)
        kk = min(k, flat.numel())
        sc, ind = topk_desc(flat, kk)
        cls = torch.div(ind, h * w, rounding_mode='trunc').int()
        pix = ind % (h * w)
        ys = torch.div(pix, w, rounding_mode='trunc').float()
        xs = (pix % w).int().float()

        def gather(t):
            return t[bi].reshape(t.shape[1], -1).t()[pix]
        ctr, cz, dm = gather(pred['center']), gather(pred['center_z']), gather(dim)
        ang = torch.atan2(gather(rot_sin), gather(rot_cos))
        xs = xs.view(-1, 1) + ctr[:, 0:1]
        ys = ys.view(-1, 1) + ctr[:, 1:2]
        xs = xs * stride * voxel_size[0] + pc_range[0]
        ys = ys * stride * voxel_size[1] + pc_range[1]
        boxes = torch.cat([xs, ys, cz, dm, ang], dim=-1)
        mask = (boxes[:, :3] >= lim[:3]).all(1) & (boxes[:, :3] <= lim[3:]).all(1)
        if score_thresh is not None:
            mask &= sc > score_thresh
        res.append({'pred_boxes': boxes[mask], 'pred_scores': sc[mask], 'pred_labels': cls[mask]})
    return res


def class_agnostic_nms(scores, boxes, thresh, pre_max, post_max):
    """model_nms_utils.py:6-25 with nms_gpu: topk(pre_max) -> sort desc -> rotated NMS -> first post_max."""
    if scores.shape[0] == 0:
        return torch.zeros((0,), dtype=torch.int64)
    kk = min(pre_max, scores.shape[0])
    sc, idx = topk_desc(scores, kk)
    b = boxes[idx][:, :7].numpy()
    keep = cref.nms_sorted(b, thresh)
    return idx[torch.from_numpy(keep[:post_max])]


def generate_predicted_boxes(pred, pc_range, voxel_size, stride, post_cfg):
    """center_head.py:315-368 (single head, labels returned 1-based)."""
    dec = decode(pred, pc_range, voxel_size, stride, post_cfg['MAX_OBJ_PER_SAMPLE'], post_cfg['SCORE_THRESH'],
                 post_cfg['POST_CENTER_LIMIT_RANGE'])
    out = []
    for d in dec:
        sel = class_agnostic_nms(d['pred_scores'], d['pred_boxes'], post_cfg['NMS_THRESH'],
                                 post_cfg['NMS_PRE_MAXSIZE'], post_cfg['NMS_POST_MAXSIZE'])
        out.append({'pred_boxes': d['pred_boxes'][sel], 'pred_scores': d['pred_scores'][sel],
                    'pred_labels': d['pred_labels'][sel].long() + 1})
    return out
