"""Function names of /root/reference/utils/detzero_utils/ops/iou3d_nms/iou3d_nms_utils.py on the HIP
backend (seam 3 of SURVEY.md §8b).  Device tensors in, device tensors out, no D2H mask copy."""
import torch

from . import ops


def boxes_overlap_bev_gpu(boxes_a, boxes_b):
    """iou3d_nms_cuda.boxes_overlap_bev_gpu: (N,7),(M,7) -> (N,M) overlap areas."""
    return ops.boxes_pairwise(boxes_a[:, :7].float().contiguous(), boxes_b[:, :7].float().contiguous(), iou=False)


def boxes_iou_bev(boxes_a, boxes_b):
    """iou3d_nms_utils.py:60-72."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    return ops.boxes_pairwise(boxes_a.float().contiguous(), boxes_b.float().contiguous(), iou=True)


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:74-107: BEV overlap x height overlap / union volume."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).reshape(-1, 1)
    a_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).reshape(-1, 1)
    b_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).reshape(1, -1)
    b_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).reshape(1, -1)
    overlaps_bev = boxes_overlap_bev_gpu(boxes_a, boxes_b)
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).reshape(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).reshape(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """iou3d_nms_utils.py:154-170.  Returns (kept indices into `boxes`, None)."""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    b = boxes[order].float().contiguous()
    if b.shape[0] == 0:
        return order, None
    keep, d_nk = ops.nms_rotated_nosync(b, None, thresh, b.shape[0])
    nk = int(d_nk.item())
    return order[keep[:nk].long()].contiguous(), None
