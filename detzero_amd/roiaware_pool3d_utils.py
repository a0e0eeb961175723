"""points_in_boxes_gpu_v2 of /root/reference/utils/detzero_utils/ops/roiaware_pool3d/
roiaware_pool3d_utils.py:45-58 on the HIP backend (refiner object crop, daemon/prepare_object_data.py:250-273)."""
from . import ops


def points_in_boxes_gpu_v2(points, boxes):
    """points (B,M,3), boxes (B,T,7) -> (B,T,M) int32, 1 = inside."""
    assert boxes.shape[0] == points.shape[0]
    assert boxes.shape[2] == 7 and points.shape[2] == 3
    return ops.points_in_boxes_v2(points.float().contiguous(), boxes.float().contiguous())
