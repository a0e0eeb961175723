"""Seeded synthetic Waymo-like LiDAR frames and detector weights.

There is no Waymo data (and no network) where this code is built and benchmarked, so every
parity test and bench run draws its inputs from here.  Shapes follow the reference's on-disk frame
format (`/root/reference/detection/detzero_det/datasets/waymo/waymo_utils.py:298-300`):
float32 rows ``[x, y, z, intensity, elongation]`` (the NLZ column is already filtered and
intensity has passed through ``tanh`` as `dataset.py:181-183` does).

Nothing here touches the GPU.
"""
import math

import numpy as np

POINT_CLOUD_RANGE = np.array([-75.2, -75.2, -2.0, 75.2, 75.2, 4.0], dtype=np.float32)
VOXEL_SIZE_01 = [0.1, 0.1, 0.15]      # BASELINE.json configs[1]  -> grid 1504 x 1504 x 40
# BASELINE.json configs[0] (20k points, 0.2 m voxels).  z stays 0.15 m: with the 0.3 m of BASELINE.md's
# sketch the grid is 20 cells high, stage 4 of VoxelResBackBone8x is 2 cells high and its (3,1,1) conv_out
# has no output at all - the reference network needs >= 33 z cells.  -> grid 752 x 752 x 40
VOXEL_SIZE_02 = [0.2, 0.2, 0.15]


def synth_waymo_frame(seed, n_points=160_000, n_beams=64, n_objects=40):
    """One spinning-LiDAR sweep: flat ground + per-azimuth obstacles + box-shaped clusters.

    Returns float32 (n_points, 5) ``[x, y, z, intensity, elongation]`` in the DetZero lidar frame
    (sensor 2 m above the ground plane, ground at z ~= 0).
    """
    rng = np.random.default_rng(seed)
    n_az = n_points // n_beams
    n_scan = n_az * n_beams
    elev = np.linspace(math.radians(-17.6), math.radians(2.4), n_beams)
    az = np.linspace(-math.pi, math.pi, n_az, endpoint=False)
    az = az + rng.uniform(0, 2 * math.pi / n_az)
    sensor_h = 2.0

    # per-azimuth obstacle range, piecewise-smooth (walls / vehicles rows)
    coarse = rng.uniform(8.0, 75.0, size=n_az // 16 + 2)
    obstacle = np.repeat(coarse, 16)[:n_az]
    el, a = np.meshgrid(elev, az, indexing='ij')            # (beams, az)
    with np.errstate(divide='ignore'):
        ground_r = np.where(el < -1e-3, sensor_h / np.tan(-el), np.inf)
    r = np.minimum(np.minimum(ground_r, obstacle[None, :] / np.cos(el)), 75.0 / np.cos(el))
    r = r + rng.normal(0.0, 0.02, size=r.shape)
    x = r * np.cos(el) * np.cos(a)
    y = r * np.cos(el) * np.sin(a)
    z = sensor_h + r * np.sin(el)
    scan = np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1)

    # replace a slice of points by box-shaped clusters so heat-maps are not degenerate
    n_obj_pts = min(n_scan // 8, n_objects * 400)
    per = max(n_obj_pts // max(n_objects, 1), 1)
    clusters = []
    for _ in range(n_objects):
        c = np.array([rng.uniform(-60, 60), rng.uniform(-60, 60), rng.uniform(0.6, 1.2)])
        dims = np.array([rng.uniform(1.5, 5.0), rng.uniform(0.6, 2.2), rng.uniform(1.2, 2.0)])
        yaw = rng.uniform(-math.pi, math.pi)
        face = rng.integers(0, 2, size=per)
        u = rng.uniform(-0.5, 0.5, size=(per, 3)) * dims
        u[face == 0, 0] = np.sign(u[face == 0, 0]) * dims[0] / 2
        u[face == 1, 1] = np.sign(u[face == 1, 1]) * dims[1] / 2
        cy, sy = math.cos(yaw), math.sin(yaw)
        px = u[:, 0] * cy - u[:, 1] * sy + c[0]
        py = u[:, 0] * sy + u[:, 1] * cy + c[1]
        pz = u[:, 2] + c[2]
        clusters.append(np.stack([px, py, pz], axis=1))
    if clusters:
        cl = np.concatenate(clusters, axis=0)
        slots = rng.choice(n_scan, size=cl.shape[0], replace=False)
        scan[slots] = cl

    pts = np.zeros((n_points, 5), dtype=np.float32)
    pts[:n_scan, :3] = scan.astype(np.float32)
    if n_scan < n_points:                                  # top up with ground returns
        extra = n_points - n_scan
        rr = rng.uniform(3, 70, size=extra)
        aa = rng.uniform(-math.pi, math.pi, size=extra)
        pts[n_scan:, 0] = rr * np.cos(aa)
        pts[n_scan:, 1] = rr * np.sin(aa)
        pts[n_scan:, 2] = rng.normal(0, 0.02, size=extra)
    pts[:, 3] = np.tanh(rng.uniform(0, 1, size=n_points)).astype(np.float32)
    pts[:, 4] = rng.uniform(0, 1, size=n_points).astype(np.float32)
    return pts


def synth_boxes(seed, n, xy_range=70.0, near_duplicates=0.3):
    """(n,7) float32 boxes ``[x,y,z,dx,dy,dz,heading]`` with clusters of near-duplicates so that
    rotated NMS has work to do (IoUs on both sides of the 0.7 threshold)."""
    rng = np.random.default_rng(seed)
    n_base = max(1, int(n * (1 - near_duplicates)))
    base = np.zeros((n_base, 7), dtype=np.float64)
    base[:, 0:2] = rng.uniform(-xy_range, xy_range, size=(n_base, 2))
    base[:, 2] = rng.uniform(-1, 2, size=n_base)
    base[:, 3] = rng.uniform(0.5, 6.0, size=n_base)
    base[:, 4] = rng.uniform(0.5, 2.5, size=n_base)
    base[:, 5] = rng.uniform(1.0, 2.5, size=n_base)
    base[:, 6] = rng.uniform(-math.pi, math.pi, size=n_base)
    dup_src = rng.integers(0, n_base, size=n - n_base)
    dup = base[dup_src].copy()
    dup[:, 0:2] += rng.normal(0, 0.15, size=(n - n_base, 2))
    dup[:, 3:5] *= rng.uniform(0.9, 1.1, size=(n - n_base, 2))
    dup[:, 6] += rng.normal(0, 0.05, size=n - n_base)
    boxes = np.concatenate([base, dup], axis=0)
    boxes = boxes[rng.permutation(n)]
    return boxes.astype(np.float32)


def merge_two_sweeps(frame_a, frame_b, dt=0.1):
    """BASELINE.json configs[4] shape: two sweeps in one frame with a time-offset column
    (`/root/reference/detection/detzero_det/datasets/dataset.py:167-195`), 6 features."""
    a = np.concatenate([frame_a, np.zeros((frame_a.shape[0], 1), np.float32)], axis=1)
    b = np.concatenate([frame_b, np.full((frame_b.shape[0], 1), dt, np.float32)], axis=1)
    return np.concatenate([a, b], axis=0)


def synth_state_dict(shapes, seed=0):
    """Deterministic synthetic weights as a pure function of (key, shape, seed): lets a golden fixture
    omit the weights (the generator of the fixture and the test both call this).  `shapes`: {key: shape}.
    Conv/linear weights ~ N(0, 1/fan_in), biases small, BatchNorm/LayerNorm gains in [0.5,1.5],
    running_var in [0.5,1.5], running_mean small, counters zero."""
    import zlib

    import torch
    out = {}
    for key, shape in shapes.items():
        rng = np.random.default_rng(zlib.crc32(key.encode()) + 7919 * seed)
        shape = tuple(int(s) for s in shape)
        last = key.rsplit('.', 1)[-1]
        if last == 'num_batches_tracked':
            t = np.zeros(shape, np.int64)
        elif last == 'running_var':
            t = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif last == 'running_mean':
            t = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        elif len(shape) <= 1:
            if last == 'weight':                       # norm gains
                t = rng.uniform(0.5, 1.5, shape).astype(np.float32)
            else:                                      # biases
                t = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            t = (rng.standard_normal(shape) / np.sqrt(max(fan_in, 1))).astype(np.float32)
        out[key] = torch.from_numpy(t)
    return out


def synth_object_track(seed, n_frames, name='Vehicle', pts_lo=0, pts_hi=600, origin=(12000.0, -3400.0, 40.0)):
    """One tracked object as the crop step stores it (daemon/prepare_object_data.py:274-313): global-frame boxes (T,7)
    float64 with headings that leave [-pi, pi), scores (T,), and per frame an (n_i,4) float64 array
    [x,y,z (global), tanh(intensity)] of points scattered in the enlarged box; n_i in [pts_lo, pts_hi] with a few
    empty frames.  Global coordinates are kilometres from the origin, as in Waymo's world frame."""
    rng = np.random.default_rng(seed)
    size = {'Vehicle': (4.6, 2.0, 1.7), 'Pedestrian': (0.9, 0.8, 1.8), 'Cyclist': (1.8, 0.8, 1.7)}[name]
    size = np.asarray(size) * rng.uniform(0.9, 1.2, size=3)
    yaw0 = rng.uniform(-np.pi, np.pi)
    speed = rng.uniform(0.0, 1.5)
    boxes = np.zeros((n_frames, 7), dtype=np.float64)
    pos = np.asarray(origin, dtype=np.float64) + rng.uniform(-50, 50, size=3) * [1, 1, 0.02]
    yaw = yaw0
    pts = []
    for t in range(n_frames):
        yaw += rng.normal(0, 0.03)
        pos = pos + speed * np.array([np.cos(yaw), np.sin(yaw), 0.0]) + rng.normal(0, 0.02, size=3)
        wrap = rng.choice([0.0, 0.0, 0.0, 2 * np.pi, -2 * np.pi])
        boxes[t] = [pos[0], pos[1], pos[2], *(size * rng.uniform(0.97, 1.03, size=3)), yaw + wrap]
        n = 0 if rng.random() < 0.08 else int(rng.integers(pts_lo, pts_hi + 1))
        local = rng.uniform(-0.55, 0.55, size=(n, 3)) * boxes[t, 3:6]
        c, s = np.cos(boxes[t, 6]), np.sin(boxes[t, 6])
        xyz = np.stack([local[:, 0] * c - local[:, 1] * s, local[:, 0] * s + local[:, 1] * c, local[:, 2]], axis=1) + boxes[t, :3]
        pts.append(np.concatenate([xyz, np.tanh(rng.uniform(0, 3, size=(n, 1)))], axis=1))
    score = rng.uniform(0.05, 0.99, size=n_frames)
    return {'boxes_global': boxes, 'score': score, 'pts': pts, 'name': name}
