"""GPU (MI355X): the x-run sparse convolution (csrc/sparse_conv_x.hip: submanifold 3 x 3 x 3 convolutions at 32 / 64 / 128 channels
with each z slab's window of input rows staged once per tile) against the CPU oracle and against the gather engine, through the
C ABI; its window prepass against the rulebook itself."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
MODES = [('f16x2', 1), ('bf16x2', 2)]
TOL = {1: 2e-4, 2: 2e-3}
K3, S1, P1 = (3, 3, 3), (1, 1, 1), (1, 1, 1)


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device).contiguous()


def _level(rng, batch, shape, density, device, cap_extra=7):
    """Random level whose z slabs have very different densities (density = per-slab fill), rows in canonical order."""
    from detzero_amd import ops
    cells = shape[0] * shape[1] * shape[2]
    slab = shape[1] * shape[2]
    lin = []
    for b in range(batch):
        for z in range(shape[0]):
            d = density[(b + z) % len(density)]
            pick = np.nonzero(rng.random(slab) < d)[0]
            lin.append(b * cells + z * slab + pick)
    lin = np.unique(np.concatenate(lin))
    coords = np.stack([lin // cells, (lin % cells) // slab, (lin // shape[2]) % shape[1], lin % shape[2]], 1).astype(np.int32)
    lvl = ops.SparseLevel(batch, shape, coords.shape[0] + cap_extra, device)
    lvl.build_from_coords(_t(coords, device), want_rank=False)
    return lvl, coords


@pytest.mark.parametrize('channels', [32, 64, 128])
def test_windows_cover_the_rulebook_exactly(device, channels):
    """windows[unit][tz] = [first, first + count) is exactly the range of input rows the nine taps of slab tz reference in the unit
    (half a kernel tile; so every row map entry lies inside, and nothing wider is staged); empty slabs give count 0, except the
    centre slab."""
    from detzero_amd import ops
    from detzero_amd import lib as L
    rng = np.random.default_rng(channels)
    lvl, coords = _level(rng, 2, [5, 40, 60], (0.02, 0.6, 0.1), device)
    nbr = ops.build_windows(lvl.neighbors_to(lvl, K3, S1, P1, packed=True), lvl, channels)
    win, tr, nbr_sorted, perm = nbr.xwin
    assert tr == L.load().dz_spconv_x_tile_rows(channels, channels) and tr in (128, 256)
    m = lvl.num_active()
    tab = ops.unpack_table(nbr)[:, :m].cpu().numpy().astype(np.int64)
    nt = (lvl.cap + tr - 1) // tr
    assert win.numel() == nt * 6 + 16 and not win[nt * 6:].any()          # window words + the (idle) tile queues
    win = win[:nt * 6].view(nt, 3, 2).cpu().numpy()
    for t in range((m + tr - 1) // tr):
        for tz in range(3):
            blk = tab[9 * tz:9 * tz + 9, t * tr:(t + 1) * tr]
            v = blk[blk >= 0]
            lo, n = win[t, tz]
            if v.size:
                assert (lo, n) == (v.min(), v.max() - v.min() + 1), (t, tz)
            else:
                assert n == (1 if tz == 1 else 0)
    assert (win[(m + tr - 1) // tr:, :, 1] == 0).all()
    # the tap-set order (64 / 128 channels; at 32 the order saves less than building it costs: ops.XRUN_SORT_MIN_CHANNELS): perm is a
    # permutation of every unit's rows (live rows at the positions below m), the sorted table holds the words of row perm[position] at
    # `position`, and masks ascend in even units / descend in odd ones
    assert (perm is None) == (channels < ops.XRUN_SORT_MIN_CHANNELS) and (nbr_sorted is None) == (perm is None)
    if perm is None:
        return
    perm = perm[:m].cpu().numpy().astype(np.int64)
    packed = nbr[:, :m].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    srt = nbr_sorted[:, :m].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    assert np.array_equal(np.sort(perm), np.arange(m)) and np.array_equal(perm // tr, np.arange(m) // tr)
    assert np.array_equal(srt, packed[:, perm])
    mask = np.zeros(m, np.int64)
    for g in range(9):
        mask |= (srt[g] >> 29) << (3 * g)
    for u in range((m + tr - 1) // tr):
        d = np.diff(mask[u * tr:(u + 1) * tr])
        assert (d <= 0).all() if u & 1 else (d >= 0).all(), u


@pytest.mark.parametrize('channels', [32, 64, 128])
def test_fused_table_and_windows_equal_the_two_launches(device, channels):
    """dz_build_neighbors_packed_x (round 5: table, tap masks, windows and tap-set order from ONE launch, the detector's route)
    against dz_build_neighbors_packed + dz_spconv_x_windows, bit for bit: levels with dense / nearly empty slabs, a capacity beyond
    the live rows (dead units, a half-live last block), a single row, and the level-2 index of a 160k-point frame."""
    from detzero_amd import ops
    from detzero_amd.synth import POINT_CLOUD_RANGE, VOXEL_SIZE_01, synth_waymo_frame
    rng = np.random.default_rng(7 * channels)
    levels = [_level(rng, 2, [5, 40, 60], (0.02, 0.6, 0.1), device)[0], _level(rng, 1, [4, 48, 64], (0.01, 0.9, 0.02, 0.5), device, cap_extra=700)[0],
              _level(rng, 1, [3, 9, 11], (0.004,), device, cap_extra=0)[0], _level(rng, 3, [3, 20, 33], (0.08,), device, cap_extra=300)[0]]
    pts = torch.from_numpy(synth_waymo_frame(3, 160000)).to(device)
    _, zyx, _, dn = ops.voxelize_hard_nosync(pts, POINT_CLOUD_RANGE, VOXEL_SIZE_01, 5, 200000, xy_range_mask=True)
    n = int(dn.item())
    c4 = torch.cat([zyx.new_zeros((n, 1)), zyx[:n]], 1).int().contiguous()
    l1 = ops.SparseLevel(1, [41, 1504, 1504], n, device)
    l1.build_from_coords(c4, want_rank=False)
    levels.append(l1.downsample(K3, (2, 2, 2), P1))
    assert ops.FUSED_XWIN
    for lvl in levels:
        two = ops.build_windows(lvl.neighbors_to(lvl, K3, S1, P1, packed=True), lvl, channels)
        one = ops.neighbors_xrun(lvl, channels)
        m = lvl.num_active()
        assert one.packed and one.kvol == 27 and one.xwin[1] == two.xwin[1]
        assert torch.equal(one[:, :m], two[:, :m]) and torch.equal(one.tile_masks, two.tile_masks)
        assert torch.equal(one.xwin[0], two.xwin[0])
        assert (one.xwin[2] is None) == (two.xwin[2] is None)
        if one.xwin[2] is not None:
            # (positions past the capacity's last whole unit are never written by either route)
            assert torch.equal(one.xwin[3][:m], two.xwin[3][:m]) and torch.equal(one.xwin[2][:, :m], two.xwin[2][:, :m])


def _run_case(device, rng, lvl, coords, channels, mid, with_res, relu, expect_gather=None):
    from detzero_amd import lib as L
    from detzero_amd import ops
    from oracle import sparse as osp
    m = coords.shape[0]
    feats = rng.standard_normal((m, channels)).astype(np.float32)
    w = (rng.standard_normal((27, channels, channels)) / np.sqrt(channels * 8)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, channels).astype(np.float32)
    shift = (rng.standard_normal(channels) * 0.1).astype(np.float32)
    res = rng.standard_normal((m, channels)).astype(np.float32)
    rb = osp.build_rulebook(coords, lvl.shape, coords, K3, S1, P1)
    ref = osp.sparse_conv(torch.from_numpy(feats), rb, torch.from_numpy(w), m)
    ref = ref * torch.from_numpy(scale) + torch.from_numpy(shift)
    if with_res:
        ref = ref + torch.from_numpy(res)
    if relu:
        ref = torch.relu(ref)
    pad = lambda a: np.concatenate([a, np.zeros((lvl.cap - m, a.shape[1]), np.float32)], 0)
    x = ops.pair16_from_f32(_t(pad(feats), device), channels, mid)
    r = ops.pair16_from_f32(_t(pad(res), device), channels, mid) if with_res else None
    ws = ops.pack_weight_split(_t(w, device), mid)
    plain = lvl.neighbors_to(lvl, K3, S1, P1)
    xt = ops.build_windows(lvl.neighbors_to(lvl, K3, S1, P1, packed=True), lvl, channels)
    assert getattr(xt, 'xwin', None) is not None
    if expect_gather is not None:
        # the windows of a tile are the union of its two units': the longest unit window is a lower bound.  Without this the
        # gather-mode arm of the kernel (windows beyond the LDS staging capacity) could go untested silently if _level changed
        rcap = L.load().dz_spconv_x_window_rows(channels, channels)
        longest = int(xt.xwin[0][:-16].view(-1, 3, 2)[..., 1].max().item())
        assert rcap > 0 and (longest > rcap) == expect_gather, (channels, longest, rcap)
    a = ops.spconv_forward(x, plain, lvl, ws, _t(scale, device), _t(shift, device), r, relu=relu, math=mid)
    b = ops.spconv_forward(x, xt, lvl, ws, _t(scale, device), _t(shift, device), r, relu=relu, math=mid)
    ga, gb = ops.pair16_to_f32(a, mid)[:m].cpu(), ops.pair16_to_f32(b, mid)[:m].cpu()
    torch.testing.assert_close(gb, ref, rtol=TOL[mid], atol=TOL[mid])
    # against the gather engine: the same products in another summation order
    torch.testing.assert_close(gb, ga, rtol=TOL[mid] / 4, atol=TOL[mid] / 4)
    return float((gb - ref).abs().max())


@pytest.mark.parametrize('name,mid', MODES)
@pytest.mark.parametrize('channels', [32, 64, 128])
def test_xrun_vs_oracle_and_gather(device, channels, name, mid):
    """Levels with dense and nearly empty z slabs side by side (windows of a tile many times its own rows: several passes), a
    batch of two, rows that are not a multiple of the tile, with / without residual and ReLU."""
    rng = np.random.default_rng(100 * channels + mid)
    # (shape, per-slab density): the second one puts a 90 %-full slab next to 1 %-full ones -> windows far beyond the LDS capacity
    # (third column: the case must / must not contain a window beyond the kernel's staging capacity = a gather-mode stage)
    for shape, dens, batch, gather in (([6, 36, 50], (0.3, 0.35, 0.25), 2, None), ([4, 48, 64], (0.01, 0.9, 0.02, 0.5), 1, True),
                                       ([3, 20, 33], (0.08,), 1, False)):
        lvl, coords = _level(rng, batch, shape, dens, device)
        for with_res, relu in ((True, True), (False, False)):
            _run_case(device, rng, lvl, coords, channels, mid, with_res, relu, expect_gather=gather)


def test_xrun_single_product_mode(device):
    """math 'f16' (one MFMA per product) runs on the same kernel; tolerance of that mode (tests/test_gpu_f16.py)."""
    from detzero_amd import ops
    rng = np.random.default_rng(9)
    lvl, coords = _level(rng, 1, [5, 30, 40], (0.3,), device)
    m = coords.shape[0]
    c = 64
    feats = rng.standard_normal((lvl.cap, c)).astype(np.float32)
    w = (rng.standard_normal((27, c, c)) / np.sqrt(c * 8)).astype(np.float32)
    x = ops.pair16_from_f32(_t(feats, device), c, 1)
    ws = ops.pack_weight_split(_t(w, device), 1)
    one = torch.ones(c, device=device)
    zero = torch.zeros(c, device=device)
    plain = lvl.neighbors_to(lvl, K3, S1, P1)
    xt = ops.build_windows(lvl.neighbors_to(lvl, K3, S1, P1, packed=True), lvl, c)
    a = ops.pair16_to_f32(ops.spconv_forward(x, plain, lvl, ws, one, zero, None, relu=False, math=3), 1)[:m]
    b = ops.pair16_to_f32(ops.spconv_forward(x, xt, lvl, ws, one, zero, None, relu=False, math=3), 1)[:m]
    torch.testing.assert_close(b, a, rtol=1e-4, atol=1e-4)
    full = ops.pair16_to_f32(ops.spconv_forward(x, xt, lvl, ws, one, zero, None, relu=False, math=1), 1)[:m]
    assert 1e-5 < float((b - full).abs().max()) < 2e-2        # really the single-product arithmetic, and in its error class


def test_xrun_refuses_what_it_does_not_cover(device):
    from detzero_amd import lib as L
    from detzero_amd import ops
    lib = L.load()
    assert lib.dz_spconv_x_tile_rows(16, 16) == 0 and lib.dz_spconv_x_tile_rows(32, 64) == 0
    rng = np.random.default_rng(1)
    lvl, coords = _level(rng, 1, [3, 16, 16], (0.3,), device)
    packed = lvl.neighbors_to(lvl, K3, S1, P1, packed=True)
    assert getattr(ops.build_windows(packed, lvl, 16), 'xwin', None) is None
    plain = lvl.neighbors_to(lvl, K3, S1, P1)
    assert getattr(ops.build_windows(plain, lvl, 64), 'xwin', None) is None        # unpacked table: left alone
    win = torch.zeros((6 + 16,), dtype=torch.int32, device=device)
    x = torch.zeros((lvl.cap, 32), device=device)
    rc = lib.dz_spconv_forward_split_x(L.ptr(x), lvl.cap, 32, L.ptr(packed), None, L.ptr(win), 128, lvl.cap, L.ptr(lvl.d_m), L.ptr(x), None, None,
                                       None, 0, L.ptr(x), 32, 1, L.stream())
    assert rc != 0 and b'tiles' in lib.dz_last_error()


@pytest.mark.parametrize('name,mid', MODES)
def test_detector_on_the_xrun_engine_160k(device, name, mid):
    """The whole detector at the headline configuration (160k points, 0.1 m voxels) with the backbone on the x-run engine: boxes within
    1e-3 of the CPU oracle, and the backbone features within the per-stage tolerances of tests/test_gpu_full_parity.py of the gather
    engine's."""
    from detzero_amd.centerpoint import FramePipeline, set_sparse_engine
    from detzero_amd.synth import VOXEL_SIZE_01
    from tests.util import cpu_state_dict, make_model, masked_frame, match_boxes, oracle_detect
    model, cfg, info = make_model(VOXEL_SIZE_01, seed=0)
    sd = cpu_state_dict(model)
    pts = masked_frame(0, 160000)
    rb = oracle_detect(sd, pts, info)['final'][0]
    n_ref = rb['pred_boxes'].shape[0]
    model = model.to(device)
    outs = {}
    for eng in ('gather', 'xrun'):
        set_sparse_engine(model, eng)
        out, d_n = FramePipeline(model, info, math=name)(torch.from_numpy(pts).to(device))
        k = int(d_n.item())
        outs[eng] = out[:k].cpu()
        nm, worst = match_boxes(rb['pred_boxes'].numpy(), rb['pred_scores'].numpy(), out[:k, :7].cpu().numpy(), out[:k, 7].cpu().numpy(), tol=1e-3)
        assert n_ref > 50 and abs(k - n_ref) <= 2 and nm >= n_ref - 2, (eng, k, n_ref, nm, worst)
        print('%s [%s]: %d boxes, %d/%d within 1e-3 of the oracle (worst %.2e)' % (eng, name, k, nm, n_ref, worst))
    set_sparse_engine(model, 'xrun')
