"""CenterHead target assignment (row C1 of SURVEY.md 8a): heat-map / regression targets from ``gt_boxes``.

The reference runs it on the CPU in a Python loop over objects even in eval mode (center_head.py:111-161, 202-260, called
unconditionally at :448-453); its result only feeds the losses and ``forward_ret_dict['target_dicts']``.  The mirror keeps
the host placement (it is a few hundred objects per frame and off the detection path) but replaces the per-object Python
loop bodies by array operations where order does not matter; the Gaussian splat keeps the reference's object order because
``torch.max`` accumulation is order independent.  Gaussian helpers: utils/centernet_utils.py:11-78.
"""
import numpy as np
import torch


def gaussian_radius(height, width, min_overlap=0.5):
    """centernet_utils.py:11-40 (the three CornerNet cases, smallest root)."""
    def root(a, b, c):
        return (b + (b ** 2 - 4 * a * c).sqrt()) / 2
    r1 = root(1, height + width, width * height * (1 - min_overlap) / (1 + min_overlap))
    r2 = root(4, 2 * (height + width), (1 - min_overlap) * width * height)
    r3 = root(4 * min_overlap, -2 * min_overlap * (height + width), (min_overlap - 1) * width * height)
    return torch.min(torch.min(r1, r2), r3)


def gaussian2d(radius):
    """centernet_utils.py:43-49 for a (2r+1, 2r+1) window, sigma = diameter / 6 (float64 like numpy's ogrid arithmetic)."""
    d = 2 * radius + 1
    sigma = d / 6
    ax = np.arange(-radius, radius + 1, dtype=np.float64)
    h = np.exp(-(ax[None, :] ** 2 + ax[:, None] ** 2) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def draw_gaussian_to_heatmap(heatmap, center, radius):
    """centernet_utils.py:52-78 (k = 1, no valid mask): element-wise max of the clipped Gaussian into the map, in place."""
    x, y = int(center[0]), int(center[1])
    height, width = heatmap.shape[0:2]
    left, right = min(x, radius), min(width - x, radius + 1)
    top, bottom = min(y, radius), min(height - y, radius + 1)
    if min(top + bottom, left + right) <= 0:
        return heatmap
    g = torch.from_numpy(gaussian2d(radius)[radius - top:radius + bottom, radius - left:radius + right]).float()
    view = heatmap[y - top:y + bottom, x - left:x + right]
    if min(g.shape) > 0 and min(view.shape) > 0:
        torch.max(view, g.to(heatmap.device), out=view)
    return heatmap


def assign_target_of_single_head(num_classes, gt_boxes, feature_map_size, feature_map_stride, point_cloud_range, voxel_size,
                                 num_max_objs=500, gaussian_overlap=0.1, min_radius=2):
    """center_head.py:111-161.  gt_boxes (N, 8) CPU ``[x,y,z,dx,dy,dz,heading,class(1-based in this head)]``; feature_map_size [x, y].
    Returns heatmap (C, H, W), ret_boxes (num_max_objs, 8) [dx,dy,z,log dims,cos,sin], inds (num_max_objs,) int64, mask."""
    heatmap = gt_boxes.new_zeros(num_classes, feature_map_size[1], feature_map_size[0])
    ret_boxes = gt_boxes.new_zeros((num_max_objs, gt_boxes.shape[-1] - 1 + 1))
    inds = gt_boxes.new_zeros(num_max_objs).long()
    mask = gt_boxes.new_zeros(num_max_objs).long()
    n = min(num_max_objs, gt_boxes.shape[0])
    if n == 0:
        return heatmap, ret_boxes, inds, mask
    x, y, z = gt_boxes[:, 0], gt_boxes[:, 1], gt_boxes[:, 2]
    coord_x = torch.clamp((x - point_cloud_range[0]) / voxel_size[0] / feature_map_stride, min=0, max=feature_map_size[0] - 0.5)
    coord_y = torch.clamp((y - point_cloud_range[1]) / voxel_size[1] / feature_map_stride, min=0, max=feature_map_size[1] - 0.5)
    center = torch.stack((coord_x, coord_y), dim=-1)
    center_int = center.int()
    dx = gt_boxes[:, 3] / voxel_size[0] / feature_map_stride
    dy = gt_boxes[:, 4] / voxel_size[1] / feature_map_stride
    radius = torch.clamp_min(gaussian_radius(dx, dy, min_overlap=gaussian_overlap).int(), min=min_radius)
    ok = (dx > 0) & (dy > 0) & (center_int[:, 0] >= 0) & (center_int[:, 0] <= feature_map_size[0]) & \
         (center_int[:, 1] >= 0) & (center_int[:, 1] <= feature_map_size[1])
    ok[n:] = False
    sel = torch.nonzero(ok).flatten()
    for k in sel.tolist():                                   # the splat itself (max-accumulation: order independent)
        draw_gaussian_to_heatmap(heatmap[int(gt_boxes[k, -1]) - 1], center[k], int(radius[k]))
    inds[sel] = (center_int[sel, 1] * feature_map_size[0] + center_int[sel, 0]).long()
    mask[sel] = 1
    ret_boxes[sel, 0:2] = center[sel] - center_int[sel].float()
    ret_boxes[sel, 2] = z[sel]
    ret_boxes[sel, 3:6] = gt_boxes[sel, 3:6].log()
    ret_boxes[sel, 6] = torch.cos(gt_boxes[sel, 6])
    ret_boxes[sel, 7] = torch.sin(gt_boxes[sel, 6])
    if gt_boxes.shape[1] > 8:
        ret_boxes[sel, 8:] = gt_boxes[sel, 7:-1]
    return heatmap, ret_boxes, inds, mask


def assign_targets(head, gt_boxes, feature_map_size):
    """center_head.py:202-260 for ``head`` (a CenterHead): gt_boxes (B, M, 8+) device or CPU tensor, feature_map_size [H, W].
    Returns {'heatmaps', 'target_boxes', 'inds', 'masks', 'heatmap_masks'} - lists with one entry per head, tensors on gt_boxes' device."""
    fms = list(feature_map_size)[::-1]                       # [H, W] -> [x, y]
    cfg = head.model_cfg.TARGET_ASSIGNER_CONFIG
    if 'vel' not in head.separate_head_cfg.HEAD_DICT.keys():
        gt_boxes = torch.cat((gt_boxes[:, :, :7], gt_boxes[:, :, -1].unsqueeze(2)), dim=2)
    dev = gt_boxes.device
    host = gt_boxes.detach().cpu()
    all_names = np.array(['bg', *head.class_names])
    ret = {'heatmaps': [], 'target_boxes': [], 'inds': [], 'masks': [], 'heatmap_masks': []}
    for cur_class_names in head.class_names_each_head:
        hm, tb, ii, mm = [], [], [], []
        for b in range(host.shape[0]):
            cur = host[b]
            names = all_names[cur[:, -1].long().numpy()]
            keep = [i for i, nme in enumerate(names) if nme in cur_class_names]
            single = cur[keep].clone() if keep else cur[:0, :]
            if keep:
                single[:, -1] = torch.tensor([cur_class_names.index(names[i]) + 1 for i in keep], dtype=single.dtype)
            out = assign_target_of_single_head(len(cur_class_names), single, fms, cfg.FEATURE_MAP_STRIDE, head.point_cloud_range,
                                               head.voxel_size, cfg.NUM_MAX_OBJS, cfg.GAUSSIAN_OVERLAP, cfg.MIN_RADIUS)
            for lst, t in zip((hm, tb, ii, mm), out):
                lst.append(t.to(dev))
        ret['heatmaps'].append(torch.stack(hm, dim=0))
        ret['target_boxes'].append(torch.stack(tb, dim=0))
        ret['inds'].append(torch.stack(ii, dim=0))
        ret['masks'].append(torch.stack(mm, dim=0))
    return ret
