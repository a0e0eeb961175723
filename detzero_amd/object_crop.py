"""Per-object point crop of the refining data path on the GPU (first half).

Mirror of the crop step of ``daemon/prepare_object_data.py:241-273,310``: for every frame of a tracked sequence the
boxes (global frame) are enlarged, the frame's points are moved to the global frame (NLZ-flagged returns dropped,
intensity through tanh) and every object receives the points inside its enlarged box.  The reference computes a dense
(T, M) mask with ``points_in_boxes_gpu_v2``, copies it to the host and boolean-indexes once per object; here
``dz_crop_points_in_boxes`` compacts on the device (bitmap mask -> bitmap scan -> one gather) and only the kept rows and
T+1 offsets come back.  The host-side preparation of the points (float64 pose product exactly as numpy does it in the
reference) is unchanged on purpose: membership is decided on the same float32 coordinates as in the reference.
"""
import numpy as np
import torch

from . import ops


def prepare_frame_points(pts, pose):
    """(N,6) [x,y,z,intensity,elongation,NLZ] lidar-frame points -> (N',4) float64 [x,y,z (global), tanh(intensity)]
    (prepare_object_data.py:262-267)."""
    pts = pts[pts[:, 5] == -1]
    homo = np.concatenate([pts[:, :3], np.ones((pts.shape[0], 1))], axis=-1) @ pose.T
    return np.concatenate([homo[:, :3], np.tanh(pts[:, 3:4])], axis=1)


def enlarge_boxes(boxes_global, enlarge_scale, crop_on_bev):
    """prepare_object_data.py:252-256"""
    b = boxes_global.copy()
    b[:, 3:6] *= enlarge_scale
    if crop_on_bev:
        b[:, 5] = 100
    return b


def crop_objects(pts_global, boxes_enlarged, device=None, cap=None):
    """pts_global (M,4) float64 (prepare_frame_points), boxes (T,7) -> list of T arrays: ``pts_global[mask[i]]`` of the
    reference, computed with one device compaction.  ``cap`` bounds the total number of kept rows (default 4 M: a point
    may fall into several overlapping boxes); a ValueError is raised when it is exceeded."""
    t, m = boxes_enlarged.shape[0], pts_global.shape[0]
    if t == 0:
        return []
    if m == 0:
        return [pts_global[:0] for _ in range(t)]
    dev = device if device is not None else torch.device('cuda', torch.cuda.current_device())
    cap = int(cap) if cap is not None else 4 * m
    xyz = torch.from_numpy(np.ascontiguousarray(pts_global[:, :3])).float().to(dev).contiguous()     # .float() as in the reference call
    payload = torch.from_numpy(np.ascontiguousarray(pts_global)).to(dev)
    boxes = torch.from_numpy(np.ascontiguousarray(boxes_enlarged[:, :7])).float().to(dev).contiguous()
    out, _, offsets, d_total = ops.crop_points_in_boxes_nosync(xyz, boxes, payload, cap)
    total = int(d_total.item())
    if total > cap:
        raise ValueError('crop_objects: %d kept points exceed the capacity %d' % (total, cap))
    rows = out[:total].cpu().numpy()
    off = offsets.cpu().numpy()
    return [rows[off[i]:off[i + 1]] for i in range(t)]


def crop_frame_objects(pts, pose, boxes_global, enlarge_scale=1.1, crop_on_bev=False, device=None):
    """One frame of prepare_object_data.py:250-273,310: raw (N,6) points + pose + (T,7) global boxes -> list of per-object points."""
    return crop_objects(prepare_frame_points(pts, pose), enlarge_boxes(np.asarray(boxes_global, dtype=np.float64), enlarge_scale, crop_on_bev),
                        device)
