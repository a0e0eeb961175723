"""CPU: the oracle's voxelizers - sequential C definition vs vectorised numpy, known answers,
and the reference's own MeanVFE / DynamicMeanVFE outputs (golden fixtures)."""
import os

import numpy as np

from detzero_amd.synth import POINT_CLOUD_RANGE, VOXEL_SIZE_01, VOXEL_SIZE_02, synth_waymo_frame
from oracle import cref
from oracle import voxelize as ov


def test_known_answer_voxel_coords():
    # SURVEY.md App. C known answers (fp32): x=75.2 -> 1503 (in grid), z=4.0 -> 40 (dropped), x=-75.2 -> 0
    pts = np.array([[75.2, 0, 0], [0, 0, 4.0], [-75.2, -75.2, -2.0]], np.float32)
    c, ok = ov.point_voxel_coords(pts, POINT_CLOUD_RANGE, VOXEL_SIZE_01)
    assert c[0, 0] == 1503 and ok[0]
    assert c[1, 2] == 40 and not ok[1]
    assert tuple(c[2]) == (0, 0, 0) and ok[2]
    assert tuple(ov.grid_size_of(POINT_CLOUD_RANGE, VOXEL_SIZE_01)) == (1504, 1504, 40)


def test_hard_voxelizer_numpy_equals_sequential_c():
    for seed, n, vs, mv in [(0, 20000, VOXEL_SIZE_02, 200000), (1, 30000, VOXEL_SIZE_01, 200000),
                            (2, 20000, VOXEL_SIZE_02, 3000), (3, 64, VOXEL_SIZE_02, 10)]:
        pts = synth_waymo_frame(seed, n)
        pts = pts[ov.mask_points_by_range(pts, POINT_CLOUD_RANGE)]
        v1, c1, n1 = cref.voxelize_hard(pts, POINT_CLOUD_RANGE, vs, 5, mv)
        v2, c2, n2 = ov.hard_voxelize(pts, POINT_CLOUD_RANGE, vs, 5, mv)
        assert np.array_equal(v1, v2) and np.array_equal(c1, c2) and np.array_equal(n1, n2)
        assert v1.shape[0] <= mv and n1.max() <= 5 and n1.min() >= 1


def test_hard_voxelizer_semantics_small():
    # two voxels, 7 points in the first: only the first 5 (input order) are kept; max_voxels=1 drops the second
    vs = [1.0, 1.0, 1.0]
    rng = [0, 0, 0, 4, 4, 4]
    pts = np.array([[0.5, 0.5, 0.5, i] for i in range(3)] + [[2.5, 0.5, 0.5, 9]] +
                   [[0.5, 0.5, 0.5, i] for i in range(3, 7)], np.float32)
    v, c, n = cref.voxelize_hard(pts, rng, vs, 5, 10)
    assert v.shape[0] == 2 and list(n) == [5, 1]
    assert list(v[0, :, 3]) == [0, 1, 2, 3, 4] and tuple(c[0]) == (0, 0, 0) and tuple(c[1]) == (0, 0, 2)
    v, c, n = cref.voxelize_hard(pts, rng, vs, 5, 1)
    assert v.shape[0] == 1 and list(n) == [5]
    v2, c2, n2 = ov.hard_voxelize(pts, rng, vs, 5, 1)
    assert np.array_equal(v, v2) and np.array_equal(c, c2)


def test_empty_inputs():
    pts = np.zeros((0, 5), np.float32)
    v, c, n = ov.hard_voxelize(pts, POINT_CLOUD_RANGE, VOXEL_SIZE_02, 5, 100)
    assert v.shape == (0, 5, 5) and c.shape == (0, 3)
    out = np.array([[500.0, 0, 0, 0, 0]], np.float32)   # all points outside the grid
    v, c, n = cref.voxelize_hard(out, POINT_CLOUD_RANGE, VOXEL_SIZE_02, 5, 100)
    assert v.shape[0] == 0


def test_meanvfe_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'det_golden.npz'))
    out = ov.mean_vfe(g['meanvfe_voxels'], g['meanvfe_num'])
    np.testing.assert_allclose(out, g['meanvfe_out'], rtol=0, atol=1e-6)


def test_dynamic_vfe_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'det_golden.npz'))
    feats, coords = ov.dynamic_mean_vfe(g['dynvfe_points'], POINT_CLOUD_RANGE, VOXEL_SIZE_02)
    assert np.array_equal(coords, g['dynvfe_coords'])          # voxel indices and order: bit-exact
    np.testing.assert_allclose(feats, g['dynvfe_feats'], rtol=1e-6, atol=1e-6)
