"""ORACLE (test infrastructure only): CPU restatement of the object crop of daemon/prepare_object_data.py:250-273,310.
The inside test is oracle/c/oracle.c's restatement of roiaware_pool3d_kernel.cu:16-36,352-374 (pinned on the reference's compiled
host twin points_in_boxes_cpu); everything else is the reference's numpy, line for line in meaning."""
import numpy as np

from . import cref


def crop_frame_objects(pts, pose, boxes_global, enlarge_scale=1.1, crop_on_bev=False):
    boxes = np.asarray(boxes_global, dtype=np.float64).copy()
    boxes[:, 3:6] *= enlarge_scale
    if crop_on_bev:
        boxes[:, 5] = 100
    pts = pts[pts[:, 5] == -1]
    g = np.concatenate([pts[:, :3], np.ones((pts.shape[0], 1))], axis=-1) @ pose.T
    pts = np.concatenate([g[:, :3], np.tanh(pts[:, 3:4])], axis=1)
    mask = cref.points_in_boxes_v2(pts[:, :3].astype(np.float32), boxes[:, :7].astype(np.float32)).astype(bool)
    return [pts[mask[i, :]] for i in range(boxes.shape[0])]
