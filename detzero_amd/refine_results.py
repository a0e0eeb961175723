"""Write-back of the refining models' outputs: object-frame predictions -> per-frame boxes in the lidar / global frame
and the per-object result records (the data format on the refiner's output side).

Mirror of ``revert_to_each_frame`` + ``generate_prediction_dicts`` of the reference's refining datasets
(waymo_geometry_dataset.py:160-250, waymo_position_dataset.py:190-286, waymo_confidence_dataset.py:164-196) and of
``box_coords_transform`` / ``world_to_lidar`` (refining/detzero_refine/utils/data_utils.py:45-57,116-124).  A few boxes
per object: host numpy like the reference (the tensors come off the device once per batch), same keys, same dtypes.
"""
import numpy as np

CLASS_NAME = {1: 'Vehicle', 2: 'Pedestrian', 3: 'Cyclist'}


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


def _yaw_matrix(yaw):
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]], dtype=np.float32)


def _wrap(angle):
    angle = np.array(angle, copy=True)
    while (angle >= np.pi).any():
        angle[angle >= np.pi] -= 2 * np.pi
    while (angle < -np.pi).any():
        angle[angle < -np.pi] += 2 * np.pi
    return angle


def box_coords_transform(traj, init_box):
    """data_utils.py:116-124: boxes in the frame of `init_box` -> global frame (in place on a copy)."""
    traj = np.array(traj, copy=True)
    traj[:, :3] = traj[:, :3] @ np.linalg.inv(_yaw_matrix(init_box[6]).T)
    traj[:, :3] += init_box[:3]
    traj[:, 6] += init_box[6]
    traj[:, 6] = _wrap(traj[:, 6])
    return traj


def world_to_lidar(boxes, poses):
    """data_utils.py:45-57: global boxes (T,7) + the frames' poses (T,4,4) -> boxes in each frame's lidar coordinates."""
    boxes, to_lidar = np.stack(boxes, axis=0), np.linalg.inv(np.stack(poses, axis=0))          # (T,7), (T,4,4) world -> lidar
    homog = np.concatenate([boxes[:, :3], np.ones((len(boxes), 1))], axis=1)
    centers = np.einsum('nk,nmk->nm', homog, to_lidar[:, :3, :])                                # row n: R_n c_n + t_n
    heading = boxes[:, 6] + np.arctan2(to_lidar[:, 1, 0], to_lidar[:, 0, 0])
    return np.concatenate([centers, boxes[:, 3:6], heading[:, None]], axis=1)


def grm_revert_to_each_frame(pred_boxes, trajectories, poses):
    """waymo_geometry_dataset.py:160-186: the refined SIZE of object i on every box of its (global) trajectory, each moved
    into its frame's lidar coordinates.  -> list of (T_i, 1, 7) arrays."""
    res = []
    for pred, traj, pose in zip(_np(pred_boxes), trajectories, poses):
        world = np.array(traj, copy=True)
        world[:, 3:6] = np.repeat(pred[3:6][None, :], len(world), axis=0)
        per_frame = []
        for ind in range(len(pose)):
            r_t = np.linalg.inv(pose[ind])
            center = np.concatenate([world[[ind], :3], np.ones((1, 1))], axis=-1) @ r_t.T
            heading = world[[ind], [6]] + np.arctan2(r_t[1, 0], r_t[0, 0])
            per_frame.append(np.concatenate([center[:, :3], world[[ind], 3:6], heading[None, :]], axis=-1))
        res.append(np.array(per_frame))
    return res


def grm_prediction_dicts(batch_dict, pred_boxes, single_pred_dict=None):
    """waymo_geometry_dataset.py:189-250.  batch_dict: sequence_name, obj_id, frame, geo_trajectory, geo_score, obj_cls, pose
    (lists per object); pred_boxes (B,7) = batch_box_preds[:, 0] of GeometryTransformer."""
    out = {} if single_pred_dict is None else single_pred_dict
    per_frame = grm_revert_to_each_frame(pred_boxes, batch_dict['geo_trajectory'], batch_dict['pose'])
    for i, boxes in enumerate(per_frame):
        seq, obj = batch_dict['sequence_name'][i], batch_dict['obj_id'][i]
        rec = {'sequence_name': seq, 'frame_id': [], 'boxes_lidar': [], 'score': [], 'name': [], 'pose': []}
        for idx, frm in enumerate(batch_dict['frame'][i]):
            rec['frame_id'].append(int(frm))
            rec['boxes_lidar'].append(boxes[idx])
            rec['score'].append(batch_dict['geo_score'][i][idx])
            rec['name'].append(CLASS_NAME[int(batch_dict['obj_cls'][i])])
            rec['pose'].append(batch_dict['pose'][i][idx])
        out.setdefault(seq, {})[obj] = rec
    return out


def prm_revert_to_each_frame(pred_boxes, init_boxes, poses):
    """waymo_position_dataset.py:259-286 (prediction branch): boxes (B, query_num, 7) in the middle box's frame ->
    (lists per object of) lidar-frame boxes (T_i,7) and global boxes (T_i,7)."""
    lidar, world = [], []
    for pred, init, pose in zip(_np(pred_boxes), _np(init_boxes), poses):
        n = len(pose)
        w = box_coords_transform(np.array(pred, copy=True), np.asarray(init))
        world.append(w[:n, :].copy())
        lidar.append(world_to_lidar(w[:n, :], pose))
    return lidar, world


def prm_prediction_dicts(batch_dict, pred_boxes, single_pred_dict=None):
    """waymo_position_dataset.py:190-257 without the ground-truth columns.  batch_dict: sequence_name, obj_id, frame,
    pos_scores, obj_cls, state, pose, pos_init_box."""
    out = {} if single_pred_dict is None else single_pred_dict
    lidar, world = prm_revert_to_each_frame(pred_boxes, batch_dict['pos_init_box'], batch_dict['pose'])
    for i in range(len(lidar)):
        seq, obj = batch_dict['sequence_name'][i], batch_dict['obj_id'][i]
        rec = {'sequence_name': seq, 'frame_id': [], 'boxes_lidar': [], 'boxes_global': [], 'score': [], 'name': [],
               'state': batch_dict['state'][i], 'pose': []}
        for idx, frm in enumerate(batch_dict['frame'][i]):
            rec['boxes_lidar'].append(lidar[i][idx])
            rec['score'].append(batch_dict['pos_scores'][i][idx])
            rec['name'].append(CLASS_NAME[int(batch_dict['obj_cls'][i])])
            rec['pose'].append(batch_dict['pose'][i][idx])
            rec['frame_id'].append(int(frm))
            rec['boxes_global'].append(world[i][idx])
        out.setdefault(seq, {})[obj] = rec
    return out


def crm_prediction_dicts(batch_dict, pred_score, single_pred_dict=None):
    """waymo_confidence_dataset.py:164-196.  batch_dict: sequence_name, obj_id, box_num, frame, conf_score (B,query_num)."""
    out = {} if single_pred_dict is None else single_pred_dict
    pred_score, conf = _np(pred_score), _np(batch_dict['conf_score'])
    for i in range(len(batch_dict['sequence_name'])):
        seq, obj, n = batch_dict['sequence_name'][i], batch_dict['obj_id'][i], int(batch_dict['box_num'][i])
        out.setdefault(seq, {})[obj] = {'sequence_name': seq, 'frame_id': np.asarray(batch_dict['frame'][i][:n]).astype(int),
                                        'score': conf[i][:n], 'new_score': pred_score[i][:n]}
    return out
