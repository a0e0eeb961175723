"""GPU (MI355X): every HIP entry point, called through the C ABI, against the CPU oracle on the
same seeded inputs and against the committed golden fixtures of the reference's own code.
Bit-exact for indices / rulebooks / integer outputs; stated tolerances for fp32."""
import os

import numpy as np
import pytest
import torch

from detzero_amd.synth import POINT_CLOUD_RANGE, VOXEL_SIZE_01, VOXEL_SIZE_02, synth_boxes, synth_waymo_frame
from tests.util import canon_order, canon_table, masked_frame

pytestmark = pytest.mark.gpu


def _t(a, device, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device).contiguous()


# ------------------------------------------------------------------------------------------------
# voxelization
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('seed,n,vs,max_voxels', [
    (0, 20000, VOXEL_SIZE_02, 200000),       # BASELINE configs[0]
    (1, 160000, VOXEL_SIZE_01, 200000),      # BASELINE configs[1]
    (2, 160000, VOXEL_SIZE_01, 30000),       # max_voxels binds: later voxels are refused
    (3, 257, VOXEL_SIZE_02, 16),
    (4, 1, VOXEL_SIZE_02, 8),
])
def test_voxelize_hard_bit_exact(device, seed, n, vs, max_voxels):
    from detzero_amd import ops
    from oracle import voxelize as ov
    pts = masked_frame(seed, n) if n > 1 else np.array([[1.0, 2.0, 0.5, 0.3, 0.1]], np.float32)
    v0, c0, n0 = ov.hard_voxelize(pts, POINT_CLOUD_RANGE, vs, 5, max_voxels)
    v1, c1, n1 = ops.voxelize_hard(_t(pts, device), POINT_CLOUD_RANGE, vs, 5, max_voxels)
    assert v1.shape[0] == v0.shape[0]
    assert np.array_equal(c1.cpu().numpy(), c0)                  # (z,y,x) indices, first-appearance order
    assert np.array_equal(n1.cpu().numpy(), n0)
    assert np.array_equal(v1.cpu().numpy(), v0)                  # point copies: bit-exact


def test_voxelize_hard_edges(device):
    from detzero_amd import ops
    from oracle import voxelize as ov
    # nothing in range / boundary values / many points in one voxel / unmasked input + xy mask flag
    out = np.array([[500.0, 0, 0, 0, 0], [0, 0, 50.0, 0, 0]], np.float32)
    v, c, n = ops.voxelize_hard(_t(out, device), POINT_CLOUD_RANGE, VOXEL_SIZE_01, 5, 100)
    assert v.shape[0] == 0
    edge = np.array([[75.2, 0, 0, 1, 1], [-75.2, -75.2, -2, 2, 2], [0, 0, 4.0, 3, 3], [0, 0, 3.9999, 4, 4],
                     [np.nextafter(np.float32(75.2), np.float32(100)), 0, 0, 5, 5]], np.float32)
    for mask in (False, True):
        pts = edge if not mask else edge
        ref_pts = pts[ov.mask_points_by_range(pts, POINT_CLOUD_RANGE)] if mask else pts
        v0, c0, n0 = ov.hard_voxelize(ref_pts, POINT_CLOUD_RANGE, VOXEL_SIZE_01, 5, 100)
        v1, c1, n1, d = ops.voxelize_hard_nosync(_t(pts, device), POINT_CLOUD_RANGE, VOXEL_SIZE_01, 5, 100, xy_range_mask=mask)
        m = int(d.item())
        assert m == v0.shape[0]
        assert np.array_equal(c1[:m].cpu().numpy(), c0) and np.array_equal(v1[:m].cpu().numpy(), v0)
    many = np.tile(np.array([[1.01, 2.02, 0.5, 0, 0]], np.float32), (300, 1))
    many[:, 3] = np.arange(300)
    v0, c0, n0 = ov.hard_voxelize(many, POINT_CLOUD_RANGE, VOXEL_SIZE_01, 5, 100)
    v1, c1, n1 = ops.voxelize_hard(_t(many, device), POINT_CLOUD_RANGE, VOXEL_SIZE_01, 5, 100)
    assert np.array_equal(v1.cpu().numpy(), v0) and list(n1.cpu().numpy()) == [5]
    empty = torch.zeros((0, 5), device=device)
    v1, c1, n1 = ops.voxelize_hard(empty, POINT_CLOUD_RANGE, VOXEL_SIZE_01, 5, 100)
    assert v1.shape[0] == 0


def test_voxelize_hard_runs_of_one_voxel(device):
    """Point orders that exercise the wave-level shortcuts of the key and insertion kernels: runs of one voxel longer than a
    wavefront (positions past max_points are dropped at once, the others start their walk at their run position), the same voxel
    in separate runs of one wavefront and of different wavefronts with SMALLER-index points arriving in later runs' neighbours,
    strictly alternating voxels (no runs), and runs that cross wavefront boundaries - against the oracle, bit for bit, through
    the list route and the level route."""
    from detzero_amd import ops
    from oracle import voxelize as ov
    rng = np.random.default_rng(11)
    cells = np.array([[762, 772, 16], [763, 772, 16], [451, 852, 20], [1152, 551, 6], [762, 773, 16]], np.float32)       # (x, y, z)
    vs = np.array(VOXEL_SIZE_01, np.float32)
    centres = np.array(POINT_CLOUD_RANGE[:3], np.float32) + (cells + 0.5) * vs
    seq = [0] * 150 + [1, 0] * 40 + [2] * 7 + [0] * 3 + [3] * 61 + [1] * 5 + [2] * 70 + [4, 3, 4, 3, 4] + [0] * 130 + [1] * 64 + [4] * 64
    seq = np.array(seq + list(rng.integers(0, 5, size=700)))
    pts = np.zeros((seq.size, 5), np.float32)
    pts[:, :3] = centres[seq] + (rng.uniform(-0.3, 0.3, size=(seq.size, 3)) * vs).astype(np.float32)
    pts[:, 3] = np.arange(seq.size)
    pts[:, 4] = rng.random(seq.size)
    v0, c0, n0 = ov.hard_voxelize(pts, POINT_CLOUD_RANGE, VOXEL_SIZE_01, 5, 100)
    assert v0.shape[0] == 5 and list(n0) == [5] * 5
    v1, c1, n1 = ops.voxelize_hard(_t(pts, device), POINT_CLOUD_RANGE, VOXEL_SIZE_01, 5, 100)
    assert np.array_equal(c1.cpu().numpy(), c0) and np.array_equal(n1.cpu().numpy(), n0) and np.array_equal(v1.cpu().numpy(), v0)
    # level route (keys with merged bitmap atomics + line flags, insertion, mean): two copies of the frame as a batch
    grid = ov.grid_size_of(POINT_CLOUD_RANGE, VOXEL_SIZE_01)
    shape = [int(grid[2]) + 1, int(grid[1]), int(grid[0])]
    both = _t(np.concatenate([pts, pts[::-1].copy()], 0), device)
    lvl, x = ops.voxelize_to_level(both, 2, POINT_CLOUD_RANGE, VOXEL_SIZE_01, 5, 200000, shape, 8, math=0, xy_range_mask=True)
    assert lvl.num_active() == 10
    got = lvl.coords[:10].cpu().numpy()
    order = np.lexsort((c0[:, 2], c0[:, 1], c0[:, 0]))
    assert np.array_equal(got[:5, 1:], c0[order]) and np.array_equal(got[5:, 1:], c0[order]) and list(got[:, 0]) == [0] * 5 + [1] * 5
    assert np.array_equal(x[:5, :5].cpu().numpy(), ov.mean_vfe(v0, n0)[order])
    vr, cr, nr = ov.hard_voxelize(pts[::-1].copy(), POINT_CLOUD_RANGE, VOXEL_SIZE_01, 5, 100)
    order_r = np.lexsort((cr[:, 2], cr[:, 1], cr[:, 0]))
    assert np.array_equal(x[5:10, :5].cpu().numpy(), ov.mean_vfe(vr, nr)[order_r])
    # dynamic route (run-merged fixed-point sums and counts): against the oracle, and bit-equal to the same points shuffled
    pb = np.concatenate([np.zeros((pts.shape[0], 1), np.float32), pts], 1)
    fd0, cd0 = ov.dynamic_mean_vfe(pb, POINT_CLOUD_RANGE, VOXEL_SIZE_01)
    fd1, cd1 = ops.voxelize_dynamic(_t(pb, device), POINT_CLOUD_RANGE, VOXEL_SIZE_01, 1)
    assert np.array_equal(cd1.cpu().numpy(), cd0)
    np.testing.assert_allclose(fd1.cpu().numpy(), fd0, rtol=1e-5, atol=1e-5)
    fd2, cd2 = ops.voxelize_dynamic(_t(pb[rng.permutation(pb.shape[0])], device), POINT_CLOUD_RANGE, VOXEL_SIZE_01, 1)
    assert torch.equal(cd2, cd1) and torch.equal(fd2, fd1)


def test_mean_vfe(device, golden_dir):
    from detzero_amd import ops
    g = np.load(os.path.join(golden_dir, 'det_golden.npz'))
    out = ops.mean_vfe(_t(g['meanvfe_voxels'], device), _t(g['meanvfe_num'], device, torch.int32))
    np.testing.assert_allclose(out.cpu().numpy(), g['meanvfe_out'], rtol=0, atol=1e-6)   # reference MeanVFE output
    out16 = ops.mean_vfe(_t(g['meanvfe_voxels'], device), _t(g['meanvfe_num'], device, torch.int32), c_out=16)
    assert torch.equal(out16[:, :5], out) and float(out16[:, 5:].abs().max()) == 0.0


def test_dynamic_vfe(device, golden_dir):
    from detzero_amd import ops
    from oracle import voxelize as ov
    g = np.load(os.path.join(golden_dir, 'det_golden.npz'))
    feats, coords = ops.voxelize_dynamic(_t(g['dynvfe_points'], device), POINT_CLOUD_RANGE, VOXEL_SIZE_02, 2)
    assert np.array_equal(coords.cpu().numpy(), g['dynvfe_coords'])                  # reference order, bit-exact
    np.testing.assert_allclose(feats.cpu().numpy(), g['dynvfe_feats'], rtol=1e-5, atol=1e-5)
    # BASELINE configs[4] shape: two merged sweeps, 6 features, 320k points
    from detzero_amd.synth import merge_two_sweeps
    pts = merge_two_sweeps(synth_waymo_frame(7, 160000), synth_waymo_frame(8, 160000))
    pb = np.concatenate([np.zeros((pts.shape[0], 1), np.float32), pts], 1)
    f0, c0 = ov.dynamic_mean_vfe(pb, POINT_CLOUD_RANGE, VOXEL_SIZE_01)
    f1, c1 = ops.voxelize_dynamic(_t(pb, device), POINT_CLOUD_RANGE, VOXEL_SIZE_01, 1)
    assert np.array_equal(c1.cpu().numpy(), c0)
    np.testing.assert_allclose(f1.cpu().numpy(), f0, rtol=1e-5, atol=1e-5)
    # deterministic: the sums are 64-bit fixed-point atomics, so the order the points land in does not matter - the same frame with
    # its points shuffled gives the same bits (fp32 atomics would not)
    perm = np.random.default_rng(0).permutation(pb.shape[0])
    f2, c2 = ops.voxelize_dynamic(_t(pb[perm], device), POINT_CLOUD_RANGE, VOXEL_SIZE_01, 1)
    assert torch.equal(c2, c1) and torch.equal(f2, f1)


# ------------------------------------------------------------------------------------------------
# sparse index / rulebook
# ------------------------------------------------------------------------------------------------
def _voxel_coords(seed, n, vs, batch=1):
    from oracle import voxelize as ov
    cs = []
    for b in range(batch):
        pts = masked_frame(seed + b, n)
        _, c, _ = ov.hard_voxelize(pts, POINT_CLOUD_RANGE, vs, 5, 200000)
        cs.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    return np.concatenate(cs, 0)


@pytest.mark.parametrize('layout', [0, 1])
@pytest.mark.parametrize('seed,n,vs,batch', [(0, 20000, VOXEL_SIZE_02, 2), (1, 160000, VOXEL_SIZE_01, 1)])
def test_index_and_rulebooks_bit_exact(device, seed, n, vs, batch, layout):
    """Active sets and all rulebooks of the backbone's index pyramid against the oracle, in both row orders: layout 0 keeps the
    rows in the canonical (linear-key) order itself; layout 1 (brick order, what the backbone runs in) is compared after sorting
    rows by the linear key - the statement of SURVEY App. C: equality of the sorted coordinate list and of the (in, out, tap) triples."""
    from detzero_amd import ops
    from oracle import sparse as osp
    from oracle import voxelize as ov
    coords = _voxel_coords(seed, n, vs, batch)
    grid = ov.grid_size_of(POINT_CLOUD_RANGE, vs)
    shape = [int(grid[2]) + 1, int(grid[1]), int(grid[0])]
    lvl = ops.SparseLevel(batch, shape, coords.shape[0], device, layout=layout)
    rank = lvl.build_from_coords(_t(coords, device))
    order = osp.canonical_order(coords, shape)
    m = lvl.num_active()
    assert m == coords.shape[0]
    got = lvl.coords[:m].cpu().numpy()
    if layout == 0:
        assert np.array_equal(got, coords[order])                                    # canonical (sorted) order
        inv = np.empty_like(order); inv[order] = np.arange(order.size)
        assert np.array_equal(rank.cpu().numpy(), inv.astype(np.int32))
    else:
        # brick order: ascending (b, y/8, x/8, z, y%8, x%8)
        c = got.astype(np.int64)
        bk = ((((c[:, 0] * ((shape[1] + 7) // 8) + c[:, 2] // 8) * ((shape[2] + 7) // 8) + c[:, 3] // 8) * shape[0] + c[:, 1]) * 64
              + (c[:, 2] % 8) * 8 + c[:, 3] % 8)
        assert np.all(np.diff(bk) > 0)
    assert np.array_equal(got[rank.cpu().numpy()], coords)                           # rank_of_input points at the input's own cell
    cur_order = canon_order(got, shape)
    assert np.array_equal(got[cur_order], coords[order])
    cur, cur_shape, cur_lvl = coords[order], shape, lvl
    K3, S1, P1 = (3, 3, 3), (1, 1, 1), (1, 1, 1)
    for k, s, p in [(K3, (2, 2, 2), (1, 1, 1)), (K3, (2, 2, 2), (1, 1, 1)), (K3, (2, 2, 2), (0, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0))]:
        # submanifold rulebook of the current level
        nbr = cur_lvl.neighbors_to(cur_lvl, K3, S1, P1)
        ref = osp.neighbor_table(cur, cur_shape, cur, K3, S1, P1)
        assert np.array_equal(canon_table(nbr[:, :cur.shape[0]].cpu().numpy(), cur_order, cur_order), ref)
        # strided conv: output set + rulebook
        nxt = cur_lvl.downsample(k, s, p)
        oc, oshape = osp.conv_out_coords(cur, cur_shape, k, s, p)
        assert nxt.shape == list(oshape) and nxt.num_active() == oc.shape[0] and nxt.layout == layout
        got_o = nxt.coords[:oc.shape[0]].cpu().numpy()
        nxt_order = canon_order(got_o, oshape)
        assert np.array_equal(got_o[nxt_order], oc)
        if layout == 0:
            assert np.array_equal(nxt_order, np.arange(oc.shape[0]))
        nbr = cur_lvl.neighbors_to(nxt, k, s, p)
        ref = osp.neighbor_table(cur, cur_shape, oc, k, s, p)
        assert np.array_equal(canon_table(nbr[:, :oc.shape[0]].cpu().numpy(), nxt_order, cur_order), ref)
        cur, cur_shape, cur_lvl, cur_order = oc, oshape, nxt, nxt_order


@pytest.mark.parametrize('shape,fill', [([5, 9, 70], 0.5), ([4, 8, 64], 0.9), ([3, 5, 33], 0.25)])
def test_rulebooks_on_dense_small_grids(device, shape, fill):
    """The per-row neighbour kernel and the wave-merged output marking where their special cases are dense: grids whose rows do not
    start on bitmap-word boundaries (x windows that straddle two words at bit 0 / 31), the first and the last word of the bitmap
    (grid corners occupied), x = 0 / W - 1, paddings that put whole (z, y) rows outside the grid - every stage shape of the
    backbone, against the oracle; the per-32-row tap masks must equal the table's own occupancy and be zero past the rows."""
    from detzero_amd import ops
    from oracle import sparse as osp
    rng = np.random.default_rng(shape[2])
    batch = 2
    cells = shape[0] * shape[1] * shape[2]
    lin = np.nonzero(rng.random(batch * cells) < fill)[0]
    lin = np.unique(np.concatenate([lin, [0, 31, 32, batch * cells - 1, batch * cells - 33]]))
    coords = np.stack([lin // cells, (lin % cells) // (shape[1] * shape[2]), (lin // shape[2]) % shape[1], lin % shape[2]], 1).astype(np.int32)
    lvl = ops.SparseLevel(batch, shape, coords.shape[0] + 5, device)
    lvl.build_from_coords(_t(coords, device), want_rank=False)
    assert lvl.num_active() == coords.shape[0]
    # (np.nonzero order = ascending linear key = the canonical order of the rows)

    def check(nbr, ref, m):
        got = nbr[:, :m].cpu().numpy()
        assert np.array_equal(got, ref)
        masks = nbr.tile_masks.cpu().numpy().astype(np.uint32)
        for gi in range(masks.shape[0]):
            blk = ref[:, gi * 32:(gi + 1) * 32]
            want = 0
            for t in range(ref.shape[0]):
                if blk.shape[1] and (blk[t] >= 0).any():
                    want |= 1 << t
            assert int(masks[gi]) == want, (gi, hex(int(masks[gi])), hex(want))

    K3, S1, P1 = (3, 3, 3), (1, 1, 1), (1, 1, 1)
    check(lvl.neighbors_to(lvl, K3, S1, P1), osp.neighbor_table(coords, shape, coords, K3, S1, P1), coords.shape[0])
    for k, s, p in [(K3, (2, 2, 2), (1, 1, 1)), (K3, (2, 2, 2), (0, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0))]:
        nxt = lvl.downsample(k, s, p)
        oc, oshape = osp.conv_out_coords(coords, shape, k, s, p)
        assert nxt.shape == list(oshape) and nxt.num_active() == oc.shape[0]
        assert np.array_equal(nxt.coords[:oc.shape[0]].cpu().numpy(), oc)
        check(lvl.neighbors_to(nxt, k, s, p), osp.neighbor_table(coords, shape, oc, k, s, p), oc.shape[0])
        check(nxt.neighbors_to(nxt, K3, S1, P1), osp.neighbor_table(oc, list(oshape), oc, K3, S1, P1), oc.shape[0])


def _random_level(rng, shape, n, batch, device, layout, dense_block=False):
    """A level of n random cells (plus, optionally, a fully occupied 12 x 32 x 32 block: tiles whose halo exceeds the LDS capacity)."""
    from detzero_amd import ops
    cells = shape[0] * shape[1] * shape[2]
    lin = rng.choice(batch * cells, size=n, replace=False)
    coords = np.stack([lin // cells, (lin % cells) // (shape[1] * shape[2]), (lin // shape[2]) % shape[1], lin % shape[2]], 1).astype(np.int32)
    if dense_block:
        z, y, x = np.meshgrid(np.arange(min(16, shape[0])), np.arange(8, 40), np.arange(8, 40), indexing='ij')
        blk = np.stack([np.zeros(z.size, np.int64), z.ravel(), y.ravel(), x.ravel()], 1).astype(np.int32)
        coords = np.unique(np.concatenate([coords, blk], 0), axis=0)
    lvl = ops.SparseLevel(batch, shape, coords.shape[0] + 7, device, layout=layout)
    lvl.build_from_coords(_t(coords, device), want_rank=False)
    return lvl


@pytest.mark.experimental
@pytest.mark.parametrize('layout', [0, 1])
@pytest.mark.parametrize('kvol', [27, 3])
def test_build_tiles_is_the_table(device, layout, kvol):
    """dz_build_tiles: per tile the halo list holds exactly the distinct neighbour rows of the tile; the slot list is the tile's
    non-empty taps in ascending order; rowmap is a permutation of the tile's rows; the local table points every (slot, sorted row)
    at its neighbour's position in the halo list (0xFFFF where the table has -1) - i.e. halo[ltab] == nbr; and the fragment slot
    masks say exactly which 32-row fragments have a neighbour at a slot."""
    from detzero_amd import lib as L
    from detzero_amd import ops
    rng = np.random.default_rng(11 + layout + kvol)
    lvl = _random_level(rng, [12, 48, 40], 2500, 2, device, layout, dense_block=True)
    if kvol == 27:
        k, s, p, out = (3, 3, 3), (1, 1, 1), (1, 1, 1), lvl
    else:
        k, s, p = (3, 1, 1), (2, 1, 1), (0, 0, 0)
        out = lvl.downsample(k, s, p)
    nbr = ops.build_tiles(lvl.neighbors_to(out, k, s, p), out)
    halo, tinfo, ltab, rowmap = (t.cpu().numpy() for t in nbr.tiles)
    tr = L.load().dz_spconv_tile_rows()
    m = out.num_active()
    tab = nbr.cpu().numpy()
    ltab = ltab.view(np.uint16)
    rowmap = rowmap.view(np.uint16).astype(np.int64)
    assert tr == 512 and m > tr and ltab.shape == ((out.cap + tr - 1) // tr, 32 * tr)
    q = np.arange(tr)
    worst = 0
    for t in range((m + tr - 1) // tr):
        n_rows = min(tr, m - t * tr)
        want = np.full((kvol, tr), -1, np.int64)
        want[:, :n_rows] = tab[:, t * tr:t * tr + n_rows]                             # (kvol, row of the tile)
        uniq = np.unique(want[want >= 0])
        nsl, nh = int(tinfo[t, 0]), int(tinfo[t, 1])
        assert nh == uniq.size and np.array_equal(np.sort(halo[t, :nh]), uniq), t   # distinct rows, each once
        taps = np.nonzero((want >= 0).any(axis=1))[0]
        assert nsl == taps.size and np.array_equal(tinfo[t, 4:4 + nsl], taps) and np.all(tinfo[t, 4 + nsl:36] == -1)
        rm = rowmap[t]
        assert np.array_equal(np.sort(rm), q)                                         # a permutation of the tile's rows
        assert np.all(rm[:n_rows] < n_rows)                                           # rows past the end sort last
        for sl in range(32):
            idx = ((((sl >> 2) * 8 + (q >> 6)) * 32 + (q & 31)) * 4 + (sl & 3)) * 2 + ((q >> 5) & 1)
            loc = ltab[t][idx].astype(np.int64)                                       # entry of (slot, sorted position q)
            if sl >= nsl:
                assert np.all(loc == 0xFFFF)
                continue
            w = want[taps[sl]][rm]                                                    # neighbour of the row at sorted position q
            assert np.array_equal(loc == 0xFFFF, w < 0)
            assert np.array_equal(np.where(loc != 0xFFFF, halo[t][np.minimum(loc, nh - 1)], -1), w), (t, sl)
            frag = (loc != 0xFFFF).reshape(16, 32).any(axis=1)
            got = (tinfo[t, 36:52].astype(np.int64) >> sl) & 1
            assert np.array_equal(got.astype(bool), frag), (t, sl)
        worst = max(worst, nh)
    assert worst > 895                                                               # the dense block needs more than one LDS pass


_TILE_SHAPES = [(16, 16, 27), (16, 32, 27), (32, 32, 27), (32, 64, 27), (64, 64, 27), (64, 128, 27), (128, 128, 27), (128, 128, 3)]
# every shape in both row orders on the default arithmetic; the other math modes run a subset (brick order, three widths)
_TILE_CASES = ([(ci, co, kv, 'f16x2', lay) for lay in (0, 1) for (ci, co, kv) in _TILE_SHAPES] +
               [(ci, co, 27, mth, 1) for mth in ('bf16x2', 'f16') for (ci, co) in ((16, 16), (64, 64), (128, 128))] +
               [(128, 128, 3, mth, 1) for mth in ('bf16x2', 'f16')])


@pytest.mark.experimental
@pytest.mark.parametrize('cin,cout,kvol,math,layout', _TILE_CASES)
def test_spconv_tiles_vs_oracle_and_gather(device, cin, cout, kvol, math, layout):
    """The tile-resident convolution against the oracle (rulebook + fp32 torch) and against the gather kernels on the same
    pair16 operands, in both row orders; the level holds a fully occupied block, so some tiles run in two LDS passes, the last
    tile is ragged, and the epilogue (BatchNorm scale / shift, residual, ReLU) is exercised with and without its parts."""
    from detzero_amd import ops
    from oracle import sparse as osp
    mid = ops.math_id(math)
    rng = np.random.default_rng(cin * 1000 + cout + kvol)
    shape = [12, 48, 40]
    lvl = _random_level(rng, shape, 2500, 2, device, layout, dense_block=True)
    n = lvl.num_active()
    coords = lvl.coords[:n].cpu().numpy()
    if kvol == 27:
        k, s, p, out_lvl = (3, 3, 3), (1, 1, 1), (1, 1, 1), lvl
    else:
        k, s, p = (3, 1, 1), (2, 1, 1), (0, 0, 0)
        out_lvl = lvl.downsample(k, s, p)
    mo = out_lvl.num_active()
    oc = out_lvl.coords[:mo].cpu().numpy()
    feats = rng.standard_normal((n, cin)).astype(np.float32)
    w = (rng.standard_normal((kvol, cin, cout)) / np.sqrt(cin * 8)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.standard_normal(cout).astype(np.float32) * 0.1
    res = rng.standard_normal((mo, cout)).astype(np.float32)
    rb = osp.build_rulebook(coords, shape, oc, k, s, p)                               # the oracle works on any row order
    ref2 = osp.sparse_conv(torch.from_numpy(feats), rb, torch.from_numpy(w), mo)
    ref = torch.relu(ref2 * torch.from_numpy(scale) + torch.from_numpy(shift) + torch.from_numpy(res))

    x = ops.pair16_from_f32(_t(np.concatenate([feats, np.zeros((lvl.cap - n, cin), np.float32)]), device), math=mid)
    wp = ops.pack_weight_split(_t(w, device), mid)
    res_pad = np.zeros((out_lvl.cap, cout), np.float32); res_pad[:mo] = res
    rp = ops.pair16_from_f32(_t(res_pad, device), math=mid)
    nbr = lvl.neighbors_to(out_lvl, k, s, p)
    g1 = ops.spconv_forward(x, nbr, out_lvl, wp, _t(scale, device), _t(shift, device), rp, relu=True, math=mid)
    g2 = ops.spconv_forward(x, nbr, out_lvl, wp, None, None, None, relu=False, math=mid, cout=cout)
    ops.build_tiles(nbr, out_lvl)
    assert nbr.tiles is not None
    t1 = ops.spconv_forward(x, nbr, out_lvl, wp, _t(scale, device), _t(shift, device), rp, relu=True, math=mid)
    t2 = ops.spconv_forward(x, nbr, out_lvl, wp, None, None, None, relu=False, math=mid, cout=cout)
    tol = {'f16x2': 2e-4, 'bf16x2': 2e-3, 'f16': 2e-2}[math]
    for got, gat, want in ((t1, g1, ref), (t2, g2, ref2)):
        a = ops.pair16_to_f32(got[:mo], mid).cpu()
        b = ops.pair16_to_f32(gat[:mo], mid).cpu()
        print('tiles %s %d->%d kvol %d layout %d: |tiles - oracle| %.2e, |tiles - gather| %.2e' % (
            math, cin, cout, kvol, layout, float((a - want).abs().max()), float((a - b).abs().max())))
        torch.testing.assert_close(a, want, rtol=tol, atol=tol)
        torch.testing.assert_close(a, b, rtol=tol, atol=tol)                          # same products, another summation order


@pytest.mark.experimental
def test_spconv_tiles_small_and_empty(device):
    """Fewer rows than a tile, a level with a single site (one tap: the step list is padded with empty taps), and no rows at all."""
    from detzero_amd import ops
    from oracle import sparse as osp
    rng = np.random.default_rng(5)
    for n in (1, 37, 0):
        shape = [5, 16, 16]
        lvl = ops.SparseLevel(1, shape, max(n, 1), device, layout=1)
        lin = rng.choice(5 * 256, size=n, replace=False)
        coords = np.stack([np.zeros(n, np.int64), lin // 256, (lin // 16) % 16, lin % 16], 1).astype(np.int32)
        lvl.build_from_coords(_t(coords.reshape(-1, 4), device), want_rank=False)
        assert lvl.num_active() == n
        cs = lvl.coords[:n].cpu().numpy()
        feats = rng.standard_normal((max(n, 1), 32)).astype(np.float32)
        w = (rng.standard_normal((27, 32, 32)) / 16).astype(np.float32)
        nbr = ops.build_tiles(lvl.neighbors_to(lvl, (3, 3, 3), (1, 1, 1), (1, 1, 1)), lvl)
        out = torch.full((lvl.cap, 32), 7.0, dtype=torch.float32, device=device)
        ops.spconv_forward(ops.pair16_from_f32(_t(feats, device), math=1), nbr, lvl, ops.pack_weight_split(_t(w, device), 1), None, None,
                           None, relu=False, out=out, math=1, cout=32)
        if n:
            rb = osp.build_rulebook(cs, shape, cs, (3, 3, 3), (1, 1, 1), (1, 1, 1))
            ref = osp.sparse_conv(torch.from_numpy(feats[:n]), rb, torch.from_numpy(w), n)
            torch.testing.assert_close(ops.pair16_to_f32(out[:n], 1).cpu(), ref, rtol=2e-4, atol=2e-4)
        else:
            assert float((out - 7.0).abs().max()) == 0                                # nothing written


def test_index_duplicates_and_empty(device):
    from detzero_amd import ops
    coords = np.array([[0, 1, 2, 3], [0, 1, 2, 3], [0, 0, 0, 0], [1, 4, 7, 7], [0, 1, 2, 2]], np.int32)
    lvl = ops.SparseLevel(2, [5, 8, 8], 8, device)
    rank = lvl.build_from_coords(_t(coords, device))
    assert lvl.num_active() == 4
    assert lvl.coords[:4].cpu().tolist() == [[0, 0, 0, 0], [0, 1, 2, 2], [0, 1, 2, 3], [1, 4, 7, 7]]
    assert rank.cpu().tolist() == [2, 2, 0, 3, 1]
    lvl0 = ops.SparseLevel(1, [5, 8, 8], 4, device)
    lvl0.build_from_coords(torch.zeros((0, 4), dtype=torch.int32, device=device), want_rank=False)
    assert lvl0.num_active() == 0
    nxt = lvl0.downsample((3, 3, 3), (2, 2, 2), (1, 1, 1))
    assert nxt.num_active() == 0


# ------------------------------------------------------------------------------------------------
# sparse convolution
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('cin,cout,kvol', [(16, 16, 27), (16, 32, 27), (32, 32, 27), (32, 64, 27), (64, 64, 27),
                                           (64, 128, 27), (128, 128, 27), (128, 128, 3)])
def test_spconv_forward_vs_oracle(device, cin, cout, kvol):
    from detzero_amd import ops
    from oracle import sparse as osp
    rng = np.random.default_rng(cin * 1000 + cout + kvol)
    shape = [9, 40, 40]
    n = 3000
    lin = rng.choice(shape[0] * shape[1] * shape[2], size=n, replace=False)
    coords = np.stack([np.zeros(n, np.int64), lin // 1600, (lin // 40) % 40, lin % 40], 1).astype(np.int32)
    coords = coords[osp.canonical_order(coords, shape)]
    feats = rng.standard_normal((n, cin)).astype(np.float32)
    if kvol == 27:
        k, s, p = (3, 3, 3), (1, 1, 1), (1, 1, 1)
        oc = coords
    else:
        k, s, p = (3, 1, 1), (2, 1, 1), (0, 0, 0)
        oc, _ = osp.conv_out_coords(coords, shape, k, s, p)
    w = (rng.standard_normal((kvol, cin, cout)) / np.sqrt(cin * 8)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.standard_normal(cout).astype(np.float32) * 0.1
    res = rng.standard_normal((oc.shape[0], cout)).astype(np.float32)
    rb = osp.build_rulebook(coords, shape, oc, k, s, p)
    ref = osp.sparse_conv(torch.from_numpy(feats), rb, torch.from_numpy(w), oc.shape[0])
    ref = torch.relu(ref * torch.from_numpy(scale) + torch.from_numpy(shift) + torch.from_numpy(res))

    lvl = ops.SparseLevel(1, shape, n, device)
    lvl.build_from_coords(_t(coords, device), want_rank=False)
    out_lvl = lvl if kvol == 27 else lvl.downsample(k, s, p)
    nbr = lvl.neighbors_to(out_lvl, k, s, p)
    res_pad = np.zeros((out_lvl.cap, cout), np.float32); res_pad[:oc.shape[0]] = res
    out = ops.spconv_forward(_t(feats, device), nbr, out_lvl, _t(w, device), _t(scale, device), _t(shift, device),
                             _t(res_pad, device), relu=True)
    torch.testing.assert_close(out[:oc.shape[0]].cpu(), ref, rtol=2e-4, atol=2e-4)
    # no epilogue: plain accumulation
    out2 = ops.spconv_forward(_t(feats, device), nbr, out_lvl, _t(w, device), None, None, None, relu=False)
    ref2 = osp.sparse_conv(torch.from_numpy(feats), rb, torch.from_numpy(w), oc.shape[0])
    torch.testing.assert_close(out2[:oc.shape[0]].cpu(), ref2, rtol=2e-4, atol=2e-4)


def test_sparse_to_bev(device):
    from detzero_amd import ops
    from oracle import sparse as osp
    rng = np.random.default_rng(3)
    shape = [2, 30, 31]
    n = 500
    lin = rng.choice(2 * shape[0] * shape[1] * shape[2], size=n, replace=False)
    cells = shape[0] * shape[1] * shape[2]
    coords = np.stack([lin // cells, (lin % cells) // (30 * 31), (lin // 31) % 30, lin % 31], 1).astype(np.int32)
    coords = coords[osp.canonical_order(coords, shape)]
    feats = rng.standard_normal((n, 128)).astype(np.float32)
    lvl = ops.SparseLevel(2, shape, n, device)
    lvl.build_from_coords(_t(coords, device), want_rank=False)
    bev = ops.sparse_to_bev(_t(feats, device), lvl, 128, pad=1)
    ref = osp.to_bev(torch.from_numpy(feats), coords, shape, 2)                 # (B, C*D, H, W)
    got = bev[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2).cpu()
    assert torch.equal(got, ref)
    assert float(bev[:, 0].abs().max()) == 0 and float(bev[:, :, -1].abs().max()) == 0   # zero border


# ------------------------------------------------------------------------------------------------
# dense convolutions: reference BaseBEVBackbone / CenterHead golden
# ------------------------------------------------------------------------------------------------
def test_bev_backbone_matches_reference_golden(device, golden_dir):
    from detzero_amd.config import AttrDict
    from detzero_amd.det_modules import BaseBEVBackbone
    g = np.load(os.path.join(golden_dir, 'det_golden.npz'))
    cfg = AttrDict({'LAYER_NUMS': [2, 2], 'LAYER_STRIDES': [1, 2], 'NUM_FILTERS': [32, 64], 'UPSAMPLE_STRIDES': [1, 2],
                    'NUM_UPSAMPLE_FILTERS': [32, 32]})
    bb = BaseBEVBackbone(cfg, 32)
    sd = {k[len('bev_backbone2d.'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('bev_backbone2d.')}
    bb.load_state_dict(sd, strict=True)
    bb = bb.to(device).eval()
    out = bb({'spatial_features': _t(g['bev_in'], device)})['spatial_features_2d']
    torch.testing.assert_close(out.cpu(), torch.from_numpy(g['bev_out']), rtol=1e-4, atol=1e-4)


def _golden_head(g, device):
    from detzero_amd.config import AttrDict
    from detzero_amd.det_modules import CenterHead
    from oracle import voxelize as ov
    hcfg = AttrDict({
        'CLASS_NAMES_EACH_HEAD': [['Vehicle', 'Pedestrian', 'Cyclist']], 'SHARED_CONV_CHANNEL': 32,
        'USE_BIAS_BEFORE_NORM': True, 'NUM_HM_CONV': 2, 'IOU_WEIGHT': 1,
        'SEPARATE_HEAD_CFG': {'HEAD_ORDER': ['center', 'center_z', 'dim', 'rot', 'iou'],
                              'HEAD_DICT': {'center': {'out_channels': 2, 'num_conv': 2}, 'center_z': {'out_channels': 1, 'num_conv': 2},
                                            'dim': {'out_channels': 3, 'num_conv': 2}, 'rot': {'out_channels': 2, 'num_conv': 2},
                                            'iou': {'out_channels': 1, 'num_conv': 2}}},
        'TARGET_ASSIGNER_CONFIG': {'FEATURE_MAP_STRIDE': 8},
        'POST_PROCESSING': {'SCORE_THRESH': 0.03, 'POST_CENTER_LIMIT_RANGE': [-80, -80, -10.0, 80, 80, 10.0],
                            'MAX_OBJ_PER_SAMPLE': 100,
                            'NMS_CONFIG': {'NMS_TYPE': 'nms_gpu', 'NMS_THRESH': 0.7, 'NMS_PRE_MAXSIZE': 4096, 'NMS_POST_MAXSIZE': 500}}})
    grid = ov.grid_size_of(POINT_CLOUD_RANGE, VOXEL_SIZE_02)
    head = CenterHead(hcfg, 64, 3, ['Vehicle', 'Pedestrian', 'Cyclist'], grid, POINT_CLOUD_RANGE, VOXEL_SIZE_02)
    sd = {k[len('head_dense_head.'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('head_dense_head.')}
    missing = head.load_state_dict(sd, strict=True)
    return head.to(device).eval()


def test_center_head_matches_reference_golden(device, golden_dir):
    from tests.util import match_boxes
    g = np.load(os.path.join(golden_dir, 'det_golden.npz'))
    head = _golden_head(g, device)
    dd = head({'spatial_features_2d': _t(g['head_in'], device), 'batch_size': 2})
    pred = head.forward_ret_dict['pred_dicts'][0]
    for name in ('center', 'center_z', 'dim', 'rot', 'iou', 'hm'):
        torch.testing.assert_close(pred[name].cpu(), torch.from_numpy(g['head_pred_' + name]), rtol=1e-4, atol=1e-4)
    for i, fb in enumerate(dd['final_box_dicts']):
        ref_b, ref_s, ref_l = g['head_boxes_%d' % i], g['head_scores_%d' % i], g['head_labels_%d' % i]
        got_b, got_s, got_l = fb['pred_boxes'].cpu().numpy(), fb['pred_scores'].cpu().numpy(), fb['pred_labels'].cpu().numpy()
        nm, worst = match_boxes(ref_b, ref_s, got_b, got_s, tol=1e-3)
        assert got_b.shape[0] == ref_b.shape[0], (got_b.shape, ref_b.shape)
        assert nm == ref_b.shape[0], (nm, ref_b.shape[0], worst)                  # boxes within 1e-3 of the reference
        assert sorted(got_l.tolist()) == sorted(ref_l.tolist())
        assert np.all(np.diff(got_s) <= 1e-7)                                      # descending scores


def test_decode_topk_matches_reference_golden(device, golden_dir):
    """Decode kernel alone, fed with the REFERENCE's head maps (so conv rounding cannot move the top-K)."""
    from detzero_amd import ops
    g = np.load(os.path.join(golden_dir, 'det_golden.npz'))
    maps = [g['head_pred_' + n] for n in ('center', 'center_z', 'dim', 'rot', 'iou', 'hm')]
    head = np.concatenate(maps, axis=1)                                   # (B,12,H,W)
    b, _, h, w = head.shape
    head_cl = np.ascontiguousarray(head.transpose(0, 2, 3, 1).reshape(b, h * w, 12))
    boxes, scores, labels, counts = ops.centerhead_decode(_t(head_cl, device), h, w, 3, 100, 0.03, [-80, -80, -10.0, 80, 80, 10.0],
                                                          POINT_CLOUD_RANGE, VOXEL_SIZE_02, 8, use_iou=True)
    for i in range(b):
        n = int(counts[i].item())
        assert n == g['dec_boxes_%d' % i].shape[0]
        assert np.array_equal(labels[i, :n].cpu().numpy(), g['dec_labels_%d' % i])          # same cells, same order
        np.testing.assert_allclose(scores[i, :n].cpu().numpy(), g['dec_scores_%d' % i], rtol=0, atol=1e-6)
        np.testing.assert_allclose(boxes[i, :n].cpu().numpy(), g['dec_boxes_%d' % i], rtol=0, atol=1e-4)


def test_topk_ties_and_small_maps(device):
    from detzero_amd import ops
    # all-equal scores: ties resolve to ascending flat index (class-major); K larger than the map
    h, w = 4, 5
    head = np.zeros((1, h * w, 12), np.float32)
    head[..., 8] = 1.0                               # iou = 1
    boxes, scores, labels, counts = ops.centerhead_decode(_t(head, device), h, w, 3, 32, 0.03, [-80, -80, -10.0, 80, 80, 10.0],
                                                          POINT_CLOUD_RANGE, VOXEL_SIZE_02, 8, use_iou=True)
    n = int(counts[0].item())
    assert n == 32
    assert labels[0, :n].cpu().tolist() == [0] * 20 + [1] * 12
    assert torch.allclose(scores[0, :n].cpu(), torch.full((n,), 0.5))
    # iou <= 0 -> score 0 -> nothing passes the 0.03 threshold
    head[..., 8] = -1.0
    _, _, _, counts = ops.centerhead_decode(_t(head, device), h, w, 3, 32, 0.03, [-80, -80, -10.0, 80, 80, 10.0],
                                            POINT_CLOUD_RANGE, VOXEL_SIZE_02, 8, use_iou=True)
    assert int(counts[0].item()) == 0


# ------------------------------------------------------------------------------------------------
# rotated IoU / NMS / points in boxes
# ------------------------------------------------------------------------------------------------
def test_rotated_iou_matches_reference_cpp_golden(device, golden_dir):
    from detzero_amd import iou3d_nms_utils
    from oracle import cref
    z = np.load(os.path.join(golden_dir, 'iou_golden.npz'))
    iou = iou3d_nms_utils.boxes_iou_bev(_t(z['a'], device), _t(z['b'], device)).cpu().numpy()
    np.testing.assert_allclose(iou, z['iou'], rtol=0, atol=2e-5)       # reference iou3d_cpu.cpp; sin/cos/atan2 are ocml vs libm
    ov = iou3d_nms_utils.boxes_overlap_bev_gpu(_t(z['a'], device), _t(z['b'], device)).cpu().numpy()
    np.testing.assert_allclose(ov, cref.boxes_overlap_bev(z['a'], z['b']), rtol=1e-5, atol=1e-4)
    assert (iou > 0.7).sum() >= 20


@pytest.mark.parametrize('n', [1, 63, 64, 65, 500, 1500])
def test_rotated_nms_vs_oracle(device, n):
    from detzero_amd import iou3d_nms_utils
    from oracle import cref
    boxes = synth_boxes(100 + n, n, xy_range=20.0 if n > 100 else 8.0, near_duplicates=0.5)
    scores = np.random.default_rng(n).permutation(n).astype(np.float32) / n          # distinct scores
    keep, _ = iou3d_nms_utils.nms_gpu(_t(boxes, device), _t(scores, device), 0.7)
    order = np.argsort(-scores, kind='stable')
    ref = order[cref.nms_sorted(boxes[order], 0.7)]
    got = keep.cpu().numpy()
    if not np.array_equal(got, ref):
        # The lists may differ only through decisions that sit ON the threshold (an IoU within float noise of 0.7), and one such
        # flip changes which later boxes are suppressed - so every decision of the device's own sweep is checked against the IoUs
        # with the boxes IT kept before: above 0.7 + eps it must have suppressed, below 0.7 - eps it must have kept
        eps = 1e-5
        iou = cref.boxes_iou_bev(boxes[order], boxes[order])
        rank = np.empty(n, np.int64); rank[order] = np.arange(n)
        kept = np.zeros(n, bool); kept[rank[got]] = True
        assert np.all(np.diff(rank[got]) > 0), 'kept boxes are not in score order'
        flips = 0
        for i in range(n):
            prev = np.nonzero(kept[:i])[0]
            worst = float(iou[i, prev].max()) if prev.size else 0.0
            if worst > 0.7 + eps:
                assert not kept[i], 'box %d kept although it overlaps a kept box by %.6f' % (i, worst)
            elif worst < 0.7 - eps:
                assert kept[i], 'box %d suppressed although its largest overlap with a kept box is %.6f' % (i, worst)
            else:
                flips += 1
        assert flips >= 1, 'NMS differs from the oracle without any on-threshold decision'
    assert len(got) < n or n < 3


def test_nms_empty(device):
    from detzero_amd import ops
    keep, d = ops.nms_rotated_nosync(torch.zeros((0, 7), device=device), None, 0.7, 500)
    assert int(d.item()) == 0
    boxes = _t(synth_boxes(1, 10), device)
    zero = torch.zeros((1,), dtype=torch.int32, device=device)
    keep, d = ops.nms_rotated_nosync(boxes, zero, 0.7, 500)
    assert int(d.item()) == 0
    keep, d = ops.nms_rotated_nosync(boxes, None, 0.7, 3)           # post_max cut
    assert int(d.item()) <= 3


def test_points_in_boxes_bit_exact(device):
    from detzero_amd import roiaware_pool3d_utils
    from oracle import cref
    boxes = synth_boxes(9, 70, xy_range=40.0)
    boxes[:, 3:6] *= 1.1                                     # daemon/prepare_object_data.py enlarges x1.1
    pts = synth_waymo_frame(9, 180000)[:, :3]
    pts = pts[(np.abs(pts[:, 0]) < 45) & (np.abs(pts[:, 1]) < 45)]
    mask = roiaware_pool3d_utils.points_in_boxes_gpu_v2(_t(pts[None], device), _t(boxes[None], device))[0].cpu().numpy()
    ref = cref.points_in_boxes_v2(pts, boxes)
    diff = int((mask != ref).sum())
    assert ref.sum() > 100
    assert diff <= 2, diff                                   # cos/sin ulp differences on the box surface only
    e = roiaware_pool3d_utils.points_in_boxes_gpu_v2(torch.zeros((1, 0, 3), device=device), _t(boxes[None], device))
    assert tuple(e.shape) == (1, 70, 0)


# ------------------------------------------------------------------------------------------------
# refiner: linear (1x1 conv) and attention core
# ------------------------------------------------------------------------------------------------
def test_linear_forward(device):
    from detzero_amd import ops
    rng = np.random.default_rng(0)
    for rows, cin, cout in [(1000, 32, 128), (4096, 128, 512), (77, 256, 256), (300, 16, 64), (513, 640, 512)]:
        x = rng.standard_normal((rows, cin)).astype(np.float32)
        w = (rng.standard_normal((cin, cout)) / np.sqrt(cin)).astype(np.float32)
        sc = rng.uniform(0.5, 1.5, cout).astype(np.float32); sh = rng.standard_normal(cout).astype(np.float32)
        y = ops.linear(_t(x, device), _t(w, device), _t(sc, device), _t(sh, device), True, cout)
        ref = np.maximum((x.astype(np.float64) @ w.astype(np.float64)) * sc + sh, 0)
        np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=2e-4, atol=2e-4)


def test_mha_core_matches_reference_golden(device, golden_dir):
    """q/k/v projections done in torch on the CPU with the reference module's weights; the fused
    attention core must reproduce the reference MultiheadAttention output."""
    from detzero_amd import ops
    g = np.load(os.path.join(golden_dir, 'mha_golden.npz'))
    q, k, v = (torch.from_numpy(g[n]) for n in ('q', 'k', 'v'))          # (L,B,E)
    w, b = torch.from_numpy(g['mha_in_proj_weight']), torch.from_numpy(g['mha_in_proj_bias'])
    e = 64
    qp = torch.nn.functional.linear(q, w[:e], b[:e]).transpose(0, 1).contiguous()
    kp = torch.nn.functional.linear(k, w[e:2 * e], b[e:2 * e]).transpose(0, 1).contiguous()
    vp = torch.nn.functional.linear(v, w[2 * e:], b[2 * e:]).transpose(0, 1).contiguous()
    o = ops.mha_core(qp.to(device), kp.to(device), vp.to(device), torch.from_numpy(g['kpm']).to(device), 2, 32 ** -0.5).cpu()
    o = torch.nn.functional.linear(o, torch.from_numpy(g['mha_out_proj.weight']), torch.from_numpy(g['mha_out_proj.bias']))
    torch.testing.assert_close(o.transpose(0, 1), torch.from_numpy(g['out']), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('b,lq,lk,heads,masked', [(3, 3, 4096, 8, False), (2, 200, 9600, 8, True), (2, 200, 200, 8, True), (1, 17, 50, 8, True),
                                                   (2, 130, 333, 8, False), (1, 33, 64, 2, True)])
def test_mha_core_vs_torch(device, b, lq, lk, heads, masked):
    from detzero_amd import ops
    gen = torch.Generator().manual_seed(lq * 7 + lk)
    e = heads * 32
    q = torch.randn((b, lq, e), generator=gen); k = torch.randn((b, lk, e), generator=gen); v = torch.randn((b, lk, e), generator=gen)
    kpm = None
    if masked:
        lens = torch.randint(5, lk + 1, (b,), generator=gen)
        kpm = torch.arange(lk)[None, :] >= lens[:, None]
    out = ops.mha_core(q.to(device), k.to(device), v.to(device), kpm.to(device) if masked else None, heads, 32 ** -0.5).cpu()
    qh = (q * 32 ** -0.5).view(b, lq, heads, 32).transpose(1, 2).double()
    kh = k.view(b, lk, heads, 32).transpose(1, 2).double()
    vh = v.view(b, lk, heads, 32).transpose(1, 2).double()
    s = qh @ kh.transpose(-1, -2)
    if masked:
        s = s.masked_fill(kpm[:, None, None, :], float('-inf'))
    ref = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(b, lq, e).float()
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('max_voxels', [200000, 3000])
def test_voxelize_hard_mean_batched_equals_per_frame(device, max_voxels):
    """One launch chain over B equally long frames == B single-frame calls, bit for bit (features, [b,z,y,x] rows in
    first-appearance order, per-frame counts), including the per-frame max_voxels cut; and the single-frame fused
    call == dz_voxelize_hard + dz_mean_vfe."""
    from detzero_amd import ops
    n, b = 30000, 3
    frames = [_t(synth_waymo_frame(40 + i, n), device) for i in range(b)]
    cap = min(max_voxels, n)
    feats, coords, d_num = ops.voxelize_hard_mean_batched(torch.cat(frames, 0), b, POINT_CLOUD_RANGE, VOXEL_SIZE_02, 5, max_voxels,
                                                          cap, xy_range_mask=True)
    for i, f in enumerate(frames):
        fi = torch.empty((cap, 5), dtype=torch.float32, device=device)
        ci = torch.full((cap, 4), -1, dtype=torch.int32, device=device)
        di = torch.zeros((1,), dtype=torch.int32, device=device)
        ops.voxelize_hard_mean_into(f, POINT_CLOUD_RANGE, VOXEL_SIZE_02, 5, max_voxels, i, fi, ci, di, xy_range_mask=True)
        m = int(di.item())
        assert m == int(d_num[i].item()) and 0 < m <= max_voxels
        assert torch.equal(coords[i * cap:i * cap + m], ci[:m]) and torch.equal(feats[i * cap:i * cap + m], fi[:m])
        assert bool((coords[i * cap + m:(i + 1) * cap, 0] == -1).all())
        # fused == unfused reference-shaped path
        vox, zyx, nump, dn = ops.voxelize_hard_nosync(f, POINT_CLOUD_RANGE, VOXEL_SIZE_02, 5, max_voxels, xy_range_mask=True)
        mean = ops.mean_vfe(vox, nump, d_m=dn)
        assert int(dn.item()) == m and torch.equal(mean[:m], fi[:m]) and torch.equal(zyx[:m], ci[:m, 1:])


@pytest.mark.parametrize('layout', [0, 1])
@pytest.mark.parametrize('math', [0, 1])
def test_voxelize_to_level_equals_voxelize_index_scatter(device, math, layout):
    """The fused batch voxelizer -> level-1 index equals hard voxelizer + MeanVFE + dz_index_from_coords + dz_scatter_rows
    bit for bit: same bitmap, prefix, canonical coordinates, count and feature rows (fp32 and pair16)."""
    from detzero_amd import ops
    n, b = 30000, 3
    frames = [_t(synth_waymo_frame(50 + i, n), device) for i in range(b)]
    shape = [41, 752, 752]                           # VOXEL_SIZE_02 grid (752,752,40) + 1 in z
    lvl, x = ops.voxelize_to_level(torch.cat(frames, 0), b, POINT_CLOUD_RANGE, VOXEL_SIZE_02, 5, 200000, shape, 16, math=math,
                                   xy_range_mask=True, layout=layout)
    feats, coords, d_num = ops.voxelize_hard_mean_batched(torch.cat(frames, 0), b, POINT_CLOUD_RANGE, VOXEL_SIZE_02, 5, 200000, n,
                                                          xy_range_mask=True)
    ref = ops.SparseLevel(b, shape, b * n, device, layout=layout)
    rank = ref.build_from_coords(coords)
    xr = ops.scatter_rows(feats, rank, 16, ref.cap, None, math=math)
    m = ref.num_active()
    assert lvl.num_active() == m == int(d_num.sum().item()) and m > 10000
    nz = lvl.bitmap != 0                      # the prefix is defined at words that hold a bit (the only ones a rank query reads)
    assert torch.equal(lvl.bitmap, ref.bitmap) and torch.equal(lvl.prefix[nz], ref.prefix[nz])
    assert torch.equal(lvl.coords[:m], ref.coords[:m])
    assert torch.equal(x[:m].view(torch.int32), xr[:m].view(torch.int32))


@pytest.mark.gpu
def test_roi_bev_features_match_reference_golden(device, golden_dir):
    """dz_roi_bev_features / CenterHead.roi_features against the reference's own get_box_center + absl_to_relative +
    bilinear_interpolate_torch + reorder_rois_for_refining_features (center_head.py:388-432,461-486; tests/golden/gen_roi_feat_golden.py):
    boxes on and over the map border included (clamped corner indices), an empty frame gives zero rows."""
    from detzero_amd import ops
    g = np.load(os.path.join(golden_dir, 'roi_feat_golden.npz'))
    bev = _t(g['bev'], device)
    boxes = _t(g['boxes0'], device)
    pcr, vs, stride = g['point_cloud_range'], g['voxel_size'], int(g['stride'])
    got = ops.roi_bev_features(boxes, bev[0].permute(1, 2, 0).contiguous(), pcr[0], pcr[1], vs[0], vs[1], stride)
    want = g['roi_features'][0]
    err = float(np.abs(got.cpu().numpy() - want).max())
    print('roi_features: max |diff| %.2e on values up to %.2f' % (err, float(np.abs(want).max())))
    assert tuple(got.shape) == want.shape and err <= 2e-5

    class Host:
        point_cloud_range, voxel_size, feature_map_stride = [float(v) for v in pcr], [float(v) for v in vs], stride
    from detzero_amd.det_modules import CenterHead
    pred = [{'pred_boxes': boxes}, {'pred_boxes': boxes[:0]}]
    full = CenterHead.roi_features(Host(), bev, pred, 37)
    assert tuple(full.shape) == g['roi_features'].shape and float(full[1].abs().max()) == 0.0
    np.testing.assert_allclose(full.cpu().numpy(), g['roi_features'], rtol=0, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('math', ['f32', 'f16x2'])
def test_two_head_center_head_matches_reference_golden(device, golden_dir, math):
    """CLASS_NAMES_EACH_HEAD with two heads ([Vehicle], [Pedestrian, Cyclist]; center_head.py:81-102, 315-385) against the reference
    class itself (tests/golden/gen_multihead_golden.py): every head's six maps, and per frame the heads' suppressed boxes one head
    after the other with the labels mapped through the head's class list."""
    import sys
    from tests.util import match_boxes
    from detzero_amd.config import AttrDict
    from detzero_amd.det_modules import CenterHead
    from detzero_amd.synth import POINT_CLOUD_RANGE, VOXEL_SIZE_02
    from oracle import voxelize as ov
    sys.path.insert(0, golden_dir)
    from gen_multihead_golden import head_cfg
    g = np.load(os.path.join(golden_dir, 'multihead_golden.npz'))
    head = CenterHead(AttrDict(head_cfg()), 64, 3, ['Vehicle', 'Pedestrian', 'Cyclist'], ov.grid_size_of(POINT_CLOUD_RANGE, VOXEL_SIZE_02),
                      POINT_CLOUD_RANGE, VOXEL_SIZE_02)
    head.load_state_dict({k[len('head_dense_head.'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('head_dense_head.')}, strict=True)
    head = head.to(device).eval().set_math(math)
    dd = head({'spatial_features_2d': _t(g['head_in'], device), 'batch_size': 2})
    worst_map = 0.0
    for i, pred in enumerate(head.forward_ret_dict['pred_dicts']):
        assert pred['hm'].shape[1] == (1, 2)[i]
        for name in ('center', 'center_z', 'dim', 'rot', 'iou', 'hm'):
            worst_map = max(worst_map, float((pred[name].cpu() - torch.from_numpy(g['pred%d_%s' % (i, name)])).abs().max()))
    assert worst_map <= 2e-5, worst_map                    # observed 1.4e-6
    for i, fb in enumerate(dd['final_box_dicts']):
        ref_b, ref_s, ref_l = g['boxes_%d' % i], g['scores_%d' % i], g['labels_%d' % i]
        got_b, got_s, got_l = fb['pred_boxes'].cpu().numpy(), fb['pred_scores'].cpu().numpy(), fb['pred_labels'].cpu().numpy()
        assert got_b.shape == ref_b.shape, (got_b.shape, ref_b.shape)
        n0 = int((ref_l == 1).sum())                                   # head 0 (Vehicle) first, then head 1
        assert np.all(got_l[:n0] == 1) and np.all(got_l[n0:] >= 2)
        nm, worst = match_boxes(ref_b, ref_s, got_b, got_s, tol=1e-3)
        assert nm == ref_b.shape[0], (nm, ref_b.shape[0], worst)
        assert sorted(got_l.tolist()) == sorted(ref_l.tolist())
    print('two-head CenterHead [%s]: maps within %.2e, %d + %d boxes' % (math, worst_map, dd['final_box_dicts'][0]['pred_boxes'].shape[0],
                                                                        dd['final_box_dicts'][1]['pred_boxes'].shape[0]))


@pytest.mark.gpu
@pytest.mark.parametrize('rows,cin,cout,relu', [(487, 41472, 256, True), (3, 4096, 19, False), (1000, 8192, 64, True)])
def test_linear_split_over_the_input_channels(device, rows, cin, cout, relu):
    """dz_linear_forward_splitk (the PDV head's 41 472 -> 256 FC layer for a few hundred RoIs, pdv_head.py:154-172) against float64 on the
    host and against dz_linear_forward: the same products, summed in eight groups."""
    from detzero_amd import ops
    rng = np.random.default_rng(rows + cin)
    x = rng.standard_normal((rows, cin)).astype(np.float32)
    cp = (cout + 15) // 16 * 16
    w = np.zeros((cin, cp), np.float32)
    w[:, :cout] = rng.standard_normal((cin, cout)).astype(np.float32) / np.sqrt(cin)
    sc, sh = rng.uniform(0.5, 1.5, cp).astype(np.float32), (0.2 * rng.standard_normal(cp)).astype(np.float32)
    assert ops.linear_splitk_ok(rows, cin)
    got = ops.linear_splitk(_t(x, device), _t(w, device), _t(sc, device), _t(sh, device), relu, cout)
    ref = ops.linear(_t(x, device), _t(w, device), _t(sc, device), _t(sh, device), relu, cout)
    want = x.astype(np.float64) @ w[:, :cout].astype(np.float64) * sc[:cout] + sh[:cout]
    if relu:
        want = np.maximum(want, 0)
    e1, e2 = float(np.abs(got.cpu().numpy() - want).max()), float(np.abs(ref.cpu().numpy() - want).max())
    print('split-k linear %d x %d -> %d: |split - f64| %.2e, |one pass - f64| %.2e' % (rows, cin, cout, e1, e2))
    assert tuple(got.shape) == (rows, cout) and e1 <= 2e-5
