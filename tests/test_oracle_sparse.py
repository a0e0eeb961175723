"""CPU: the sparse-conv oracle (spconv is un-vendored -> parity unpinned) is cross-checked against
dense torch.nn.functional.conv3d on densified inputs, as SURVEY.md §7 'hard parts' prescribes."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import sparse as osp


def _random_sparse(seed, shape, n, c, batch=2):
    rng = np.random.default_rng(seed)
    cells = batch * shape[0] * shape[1] * shape[2]
    lin = rng.choice(cells, size=n, replace=False)
    b = lin // (shape[0] * shape[1] * shape[2])
    r = lin % (shape[0] * shape[1] * shape[2])
    z, y, x = r // (shape[1] * shape[2]), (r // shape[2]) % shape[1], r % shape[2]
    coords = np.stack([b, z, y, x], 1).astype(np.int32)
    feats = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32))
    return coords, feats


def _densify(coords, feats, shape, batch):
    d = torch.zeros((batch, feats.shape[1], *shape))
    cc = torch.from_numpy(coords.astype(np.int64))
    d[cc[:, 0], :, cc[:, 1], cc[:, 2], cc[:, 3]] = feats
    return d


def test_subm_conv_equals_masked_dense_conv():
    shape, batch = (6, 9, 10), 2
    coords, feats = _random_sparse(0, shape, 150, 4, batch)
    w = torch.randn(8, 3, 3, 3, 4)                      # spconv layout (Cout,kD,kH,kW,Cin)
    rb = osp.build_rulebook(coords, shape, coords, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    out = osp.sparse_conv(feats, rb, osp.weight_to_taps(w), coords.shape[0])
    dense = F.conv3d(_densify(coords, feats, shape, batch), w.permute(0, 4, 1, 2, 3), padding=1)
    cc = torch.from_numpy(coords.astype(np.int64))
    ref = dense[cc[:, 0], :, cc[:, 1], cc[:, 2], cc[:, 3]]
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)


def _check_strided(k, s, p, shape):
    batch = 2
    coords, feats = _random_sparse(1, shape, 120, 3, batch)
    w = torch.randn(5, *k, 3)
    oc, oshape = osp.conv_out_coords(coords, shape, k, s, p)
    rb = osp.build_rulebook(coords, shape, oc, k, s, p)
    out = osp.sparse_conv(feats, rb, osp.weight_to_taps(w), oc.shape[0])
    dense = F.conv3d(_densify(coords, feats, shape, batch), w.permute(0, 4, 1, 2, 3), stride=s, padding=p)
    assert list(dense.shape[2:]) == list(oshape)
    cc = torch.from_numpy(oc.astype(np.int64))
    torch.testing.assert_close(out, dense[cc[:, 0], :, cc[:, 1], cc[:, 2], cc[:, 3]], rtol=1e-5, atol=1e-5)
    # active output set == cells whose receptive field touches >= 1 active input (ones-kernel test)
    occ = _densify(coords, torch.ones(coords.shape[0], 1), shape, batch)
    touched = F.conv3d(occ, torch.ones(1, 1, *k), stride=s, padding=p)[:, 0] > 0
    mask = torch.zeros_like(touched)
    mask[cc[:, 0], cc[:, 1], cc[:, 2], cc[:, 3]] = True
    assert torch.equal(mask, touched)
    # canonical order
    key = osp.lin_key(oc, oshape)
    assert np.all(np.diff(key) > 0)


def test_strided_conv_equals_dense_conv():
    _check_strided((3, 3, 3), (2, 2, 2), (1, 1, 1), (7, 10, 12))      # spconv2 / spconv3 geometry (odd + even extents)
    _check_strided((3, 3, 3), (2, 2, 2), (0, 1, 1), (11, 8, 8))       # spconv4: padding (0,1,1)
    _check_strided((3, 1, 1), (2, 1, 1), (0, 0, 0), (5, 6, 6))        # conv_out


def test_neighbor_table_matches_rulebook():
    shape = (5, 8, 8)
    coords, _ = _random_sparse(2, shape, 90, 1, 1)
    order = osp.canonical_order(coords, shape)
    coords = coords[order]
    i, o, t = osp.build_rulebook(coords, shape, coords, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    tab = osp.neighbor_table(coords, shape, coords, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    assert (tab >= 0).sum() == i.size
    assert np.all(tab[13] == np.arange(coords.shape[0]))               # centre tap = identity for subm
    assert np.array_equal(tab[t, o], i)


def test_backbone_shapes_and_bev():
    from detzero_amd.synth import POINT_CLOUD_RANGE, VOXEL_SIZE_02, synth_waymo_frame
    from oracle import voxelize as ov
    from detzero_amd.det_modules import VoxelResBackBone8x
    from detzero_amd.config import AttrDict
    pts = synth_waymo_frame(0, 4000)
    pts = pts[ov.mask_points_by_range(pts, POINT_CLOUD_RANGE)]
    vox, c, n = ov.hard_voxelize(pts, POINT_CLOUD_RANGE, VOXEL_SIZE_02, 5, 200000)
    feats = ov.mean_vfe(vox, n)
    coords = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
    grid = ov.grid_size_of(POINT_CLOUD_RANGE, VOXEL_SIZE_02)
    bb = VoxelResBackBone8x(AttrDict({}), 5, grid).eval()
    sd = {('backbone3d.' + k): v for k, v in bb.state_dict().items()}
    res = osp.backbone_forward(sd, feats, coords, bb.sparse_shape)
    assert res['x_conv1'][2] == [41, 752, 752] and res['x_conv2'][2] == [21, 376, 376]
    assert res['x_conv3'][2] == [11, 188, 188] and res['x_conv4'][2] == [5, 94, 94]
    x, oc, shape = res['encoded']
    assert shape == [2, 94, 94] and x.shape[1] == 128 and x.shape[0] == oc.shape[0] > 0
    bev = osp.to_bev(x, oc, shape, 1)
    assert tuple(bev.shape) == (1, 256, 94, 94)
    # channel index = c*D + d (height_compression.py:23)
    j = 0
    b, z, y, xx = oc[j]
    assert torch.equal(bev[0, z::2, y, xx], x[j])


def test_all_nine_rulebooks_of_the_backbone_against_a_dense_lookup():
    """The nine rulebooks of VoxelResBackBone8x on the 20k-point / 0.2 m configuration (backbone3d.py:243-280: subm1, res1,
    spconv2, res2, spconv3, res3, spconv4 with padding (0,1,1), res4, and the (3,1,1)/(2,1,1) conv_out), each derived a second,
    independent way: a DENSE index map (cell -> row + 1) read at out * stride - pad + tap for every output and tap, and output
    sets from a ones-kernel dense conv3d.  The oracle's searchsorted-based tables must agree entry for entry."""
    from detzero_amd.synth import POINT_CLOUD_RANGE, VOXEL_SIZE_02, synth_waymo_frame
    from oracle import voxelize as ov
    pts = synth_waymo_frame(0, 20000)
    pts = pts[ov.mask_points_by_range(pts, POINT_CLOUD_RANGE)]
    _, czyx, _ = ov.hard_voxelize(pts, POINT_CLOUD_RANGE, VOXEL_SIZE_02, 5, 200000)
    coords = np.concatenate([np.zeros((czyx.shape[0], 1), np.int32), czyx], 1)
    shape = [41, 752, 752]
    coords = coords[osp.canonical_order(coords, shape)]
    K3, S1, P1, S2 = (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 2, 2)
    plan = [('subm1', None), ('res1', None), ('spconv2', (K3, S2, (1, 1, 1))), ('res2', None), ('spconv3', (K3, S2, (1, 1, 1))), ('res3', None),
            ('spconv4', (K3, S2, (0, 1, 1))), ('res4', None), ('spconv_down2', ((3, 1, 1), (2, 1, 1), (0, 0, 0)))]

    def dense_table(cin, sin, cout, k, s, p):
        idx = np.zeros(sin, np.int32)
        idx[cin[:, 1], cin[:, 2], cin[:, 3]] = np.arange(1, cin.shape[0] + 1)
        tab = np.full((k[0] * k[1] * k[2], cout.shape[0]), -1, np.int32)
        for tz in range(k[0]):
            for ty in range(k[1]):
                for tx in range(k[2]):
                    u = cout[:, 1:].astype(np.int64) * np.array(s) - np.array(p) + np.array([tz, ty, tx])
                    ok = np.all((u >= 0) & (u < np.array(sin)), axis=1)
                    v = np.zeros(cout.shape[0], np.int32)
                    v[ok] = idx[u[ok, 0], u[ok, 1], u[ok, 2]]
                    tab[(tz * k[1] + ty) * k[2] + tx] = v - 1
        return tab
    checked = 0
    for name, geom in plan:
        if geom is None:
            k, s, p, oc, oshape = K3, S1, P1, coords, shape
        else:
            k, s, p = geom
            oc, oshape = osp.conv_out_coords(coords, shape, k, s, p)
            occ = torch.zeros((1, 1, *shape))
            occ[0, 0, coords[:, 1], coords[:, 2], coords[:, 3]] = 1
            touched = F.conv3d(occ, torch.ones(1, 1, *k), stride=s, padding=p)[0, 0] > 0
            got = torch.zeros_like(touched)
            got[oc[:, 1], oc[:, 2], oc[:, 3]] = True
            assert torch.equal(got, touched), name                       # output set = cells whose window holds an input
            assert list(touched.shape) == list(oshape)
        tab = osp.neighbor_table(coords, shape, oc, k, s, p)
        assert np.array_equal(tab, dense_table(coords, shape, oc, k, s, p)), name
        checked += 1
        coords, shape = oc, list(oshape)
    assert checked == 9 and shape == [2, 94, 94]


def test_packed_table_format_against_the_rulebook():
    """The packed form of a 27-tap table (include/detzero_hip.h: dz_build_neighbors_packed) rests on one property of rows kept in
    key order: the three x taps of a window row are CONSECUTIVE input rows.  Checked on the oracle's own tables (submanifold and
    both strided paddings of the backbone), then packed here as the header states and unpacked by the host helper."""
    from detzero_amd import ops
    shape, batch = (7, 12, 37), 2
    coords, _ = _random_sparse(3, shape, 2500, 1, batch)
    coords = coords[osp.canonical_order(coords, shape)]
    for k, s, p in (((3, 3, 3), (1, 1, 1), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1))):
        oc = coords if s == (1, 1, 1) else osp.conv_out_coords(coords, shape, k, s, p)[0]
        tab = osp.neighbor_table(coords, shape, oc, k, s, p).astype(np.int64)            # (27, M)
        t3 = tab.reshape(9, 3, -1)
        left, cen, right = t3[:, 0], t3[:, 1], t3[:, 2]
        both = (left >= 0) & (cen >= 0)
        assert np.array_equal(left[both] + 1, cen[both])
        both = (cen >= 0) & (right >= 0)
        assert np.array_equal(cen[both] + 1, right[both])
        both = (left >= 0) & (cen < 0) & (right >= 0)
        assert np.array_equal(left[both] + 1, right[both])                               # centre empty: right follows left
        r = np.where(left >= 0, left + 1, np.where(cen >= 0, cen, np.where(right >= 0, right, 0)))
        word = r | ((left >= 0).astype(np.int64) << 29) | ((cen >= 0).astype(np.int64) << 30) | ((right >= 0).astype(np.int64) << 31)
        packed = torch.from_numpy(word.astype(np.uint32).view(np.int32))
        packed.packed, packed.kvol = True, 27
        assert torch.equal(ops.unpack_table(packed), torch.from_numpy(tab.astype(np.int32)))
        assert ops.table_pairs(packed, oc.shape[0]) == int((tab >= 0).sum())
        plain = torch.from_numpy(tab.astype(np.int32))
        assert ops.unpack_table(plain) is plain and ops.table_pairs(plain, oc.shape[0]) == int((tab >= 0).sum())
