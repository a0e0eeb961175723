"""Tensor-level wrappers over the C ABI (one Python function per entry point).

Everything here is plumbing: allocate outputs/workspaces as torch tensors on the current device,
pass raw pointers + the current HIP stream to libdetzero_hip, raise on a non-zero status.
Functions whose result size is only known on the device come in two flavours: the plain one
returns tensors sliced to the exact size (one host sync, reference-compatible), ``*_nosync``
returns capacity-sized tensors plus the device count (graph-capturable).
"""
import ctypes
import os

import numpy as np
import torch

from . import lib as L


def _dev():
    return torch.device('cuda', torch.cuda.current_device())


def _ws(nbytes):
    return torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=_dev())


def grid_size_of(pc_range, voxel_size):
    """round((hi-lo)/voxel) as in data_processor.py:63-65 (float32 range, float64 voxel size)."""
    r = np.asarray(pc_range, dtype=np.float32)
    return np.round((r[3:6] - r[0:3]) / np.array(voxel_size)).astype(np.int64)


# ------------------------------------------------------------------------------------------------
# voxelization
# ------------------------------------------------------------------------------------------------
def voxelize_hard_nosync(points, pc_range, voxel_size, max_points, max_voxels, xy_range_mask=False):
    """points (N,C) f32 cuda -> voxels (max_voxels,max_points,C), coords_zyx (max_voxels,3) i32,
    num_points (max_voxels) i32, d_num (1,) i32 device count."""
    lib = L.load()
    L.require_cuda(points)
    n, c = points.shape
    grid = grid_size_of(pc_range, voxel_size)
    cap = int(min(max_voxels, max(n, 1)))
    voxels = torch.empty((cap, max_points, c), dtype=torch.float32, device=points.device)
    coords = torch.empty((cap, 3), dtype=torch.int32, device=points.device)
    nump = torch.empty((cap,), dtype=torch.int32, device=points.device)
    d_num = torch.zeros((1,), dtype=torch.int32, device=points.device)
    wsb = lib.dz_voxelize_hard_workspace_bytes(n, int(grid[0]), int(grid[1]), int(grid[2]), max_points)
    ws = _ws(wsb)
    rc = lib.dz_voxelize_hard(L.ptr(points), n, c, L.f6(pc_range), L.f3(voxel_size), L.i3(grid),
                              1 if xy_range_mask else 0, max_points, cap, L.ptr(voxels), L.ptr(coords), L.ptr(nump), L.ptr(d_num), L.ptr(ws),
                              ws.numel(), L.stream())
    L.check(rc, 'dz_voxelize_hard')
    return voxels, coords, nump, d_num


def voxelize_hard_mean_into(points, pc_range, voxel_size, max_points, max_voxels, batch_index, feats, coords, d_num,
                            xy_range_mask=False):
    """Fused hard voxelizer + MeanVFE writing into caller-owned slices: feats (cap, C') f32, coords (cap,4) i32
    [b,z,y,x] (pre-filled with -1 by the caller), d_num (1,) i32.  cap = feats.shape[0]."""
    lib = L.load()
    L.require_cuda(points, feats, coords, d_num)
    n, c = points.shape
    grid = grid_size_of(pc_range, voxel_size)
    cap = feats.shape[0]
    ws = _ws(lib.dz_voxelize_hard_workspace_bytes(n, int(grid[0]), int(grid[1]), int(grid[2]), max_points))
    rc = lib.dz_voxelize_hard_mean(L.ptr(points), n, c, L.f6(pc_range), L.f3(voxel_size), L.i3(grid), 1 if xy_range_mask else 0,
                                   max_points, int(min(max_voxels, cap)), int(batch_index), L.ptr(feats), feats.shape[1],
                                   L.ptr(coords), L.ptr(d_num), L.ptr(ws), ws.numel(), L.stream())
    L.check(rc, 'dz_voxelize_hard_mean')


def voxelize_hard_mean_batched(points, batch, pc_range, voxel_size, max_points, max_voxels, cap_per_frame, xy_range_mask=False):
    """points (batch*n, C): `batch` equally long frames back to back -> feats (batch*cap, C), coords (batch*cap,4) i32
    [b,z,y,x] with b = -1 on unused rows, d_num (batch,) i32 - one launch chain for the whole batch."""
    lib = L.load()
    L.require_cuda(points)
    n_tot, c = points.shape
    n_per = n_tot // batch
    assert n_per * batch == n_tot
    grid = grid_size_of(pc_range, voxel_size)
    dev = points.device
    feats = torch.empty((batch * cap_per_frame, c), dtype=torch.float32, device=dev)
    coords = torch.full((batch * cap_per_frame, 4), -1, dtype=torch.int32, device=dev)
    d_num = torch.zeros((batch,), dtype=torch.int32, device=dev)
    ws = _ws(lib.dz_voxelize_hard_batched_workspace_bytes(n_per, batch, int(grid[0]), int(grid[1]), int(grid[2]), max_points))
    rc = lib.dz_voxelize_hard_mean_batched(L.ptr(points), n_per, batch, c, L.f6(pc_range), L.f3(voxel_size), L.i3(grid),
                                           1 if xy_range_mask else 0, max_points, int(max_voxels), L.ptr(feats), c, L.ptr(coords),
                                           int(cap_per_frame), L.ptr(d_num), L.ptr(ws), ws.numel(), L.stream())
    L.check(rc, 'dz_voxelize_hard_mean_batched')
    return feats, coords, d_num


def voxelize_to_level(points, batch, pc_range, voxel_size, max_points, max_voxels, level_shape, c_dst, math=0, xy_range_mask=False,
                      layout=0):
    """points (batch*n, C), equally long frames back to back, n <= max_voxels -> (SparseLevel of the frames' voxels, level-1
    feature rows (cap, c_dst) fp32 or pair16): hard voxelizer + MeanVFE + sparse-tensor construction in one launch chain."""
    lib = L.load()
    L.require_cuda(points)
    n_tot, c = points.shape
    n_per = n_tot // batch
    assert n_per * batch == n_tot
    grid = grid_size_of(pc_range, voxel_size)
    assert [int(level_shape[1]), int(level_shape[2])] == [int(grid[1]), int(grid[0])], (level_shape, grid)
    cap = batch * max(min(int(max_voxels), n_per), 1)
    lvl = SparseLevel(batch, level_shape, cap, points.device, layout=layout, zero_count=False)
    feats = torch.empty((cap, c_dst), dtype=torch.float32, device=points.device)
    ws = _ws(lib.dz_voxelize_to_level_workspace_bytes(n_per, batch, max_points, cap, *lvl.shape, lvl.layout))
    rc = lib.dz_voxelize_to_level(L.ptr(points), n_per, batch, c, L.f6(pc_range), L.f3(voxel_size), L.i3(grid),
                                  1 if xy_range_mask else 0, max_points, int(max_voxels), lvl.shape[0], lvl.layout, L.ptr(lvl.bitmap),
                                  L.ptr(lvl.prefix), L.ptr(lvl.coords), L.ptr(lvl.d_m), cap, L.ptr(feats), c_dst, storage_math(math),
                                  L.ptr(ws), ws.numel(), L.stream())
    L.check(rc, 'dz_voxelize_to_level')
    lvl.prefix_partial = True
    return lvl, feats


def voxelize_hard(points, pc_range, voxel_size, max_points, max_voxels):
    voxels, coords, nump, d_num = voxelize_hard_nosync(points, pc_range, voxel_size, max_points, max_voxels)
    m = int(d_num.item())
    return voxels[:m], coords[:m], nump[:m]


def mean_vfe(voxels, num_points, c_out=None, d_m=None):
    lib = L.load()
    L.require_cuda(voxels, num_points)
    m, p, c = voxels.shape
    c_out = c if c_out is None else c_out
    out = torch.empty((m, c_out), dtype=torch.float32, device=voxels.device)
    rc = lib.dz_mean_vfe(L.ptr(voxels), L.ptr(num_points), L.ptr(d_m), m, p, c, L.ptr(out), c_out, L.stream())
    L.check(rc, 'dz_mean_vfe')
    return out


def voxelize_dynamic_nosync(points_b, pc_range, voxel_size, batch_size, cap=None, xy_range_mask=False):
    lib = L.load()
    L.require_cuda(points_b)
    n, c1 = points_b.shape
    c = c1 - 1
    grid = grid_size_of(pc_range, voxel_size)
    cap = int(max(n, 1)) if cap is None else int(cap)
    feats = torch.empty((cap, c), dtype=torch.float32, device=points_b.device)
    coords = torch.empty((cap, 4), dtype=torch.int32, device=points_b.device)
    d_num = torch.zeros((1,), dtype=torch.int32, device=points_b.device)
    wsb = lib.dz_voxelize_dynamic_workspace_bytes(n, batch_size, int(grid[0]), int(grid[1]), int(grid[2]), c, cap)
    ws = _ws(wsb)
    rc = lib.dz_voxelize_dynamic_mean(L.ptr(points_b), n, c, L.f6(pc_range), L.f3(voxel_size), L.i3(grid),
                                      1 if xy_range_mask else 0, batch_size, L.ptr(feats), L.ptr(coords), L.ptr(d_num), cap, L.ptr(ws),
                                      ws.numel(), L.stream())
    L.check(rc, 'dz_voxelize_dynamic_mean')
    return feats, coords, d_num


def voxelize_dynamic(points_b, pc_range, voxel_size, batch_size):
    feats, coords, d_num = voxelize_dynamic_nosync(points_b, pc_range, voxel_size, batch_size)
    m = int(d_num.item())
    return feats[:m], coords[:m]


# ------------------------------------------------------------------------------------------------
# sparse index
# ------------------------------------------------------------------------------------------------
LAYOUT_LINEAR, LAYOUT_BRICK = 0, 1


class SparseLevel:
    """One resolution level of the sparse tensor: bitmap + popcount prefix + coordinates of the rows.
    Plays the role of spconv's SparseConvTensor.indices / indice_dict (backbone3d.py:302-307).
    layout: the cell key that orders the rows (include/detzero_hip.h: DZ_LAYOUT_LINEAR = ascending (b, z, y, x), the canonical
    order of the parity statements; DZ_LAYOUT_BRICK = 8 x 8 columns of the (y, x) plane, the order the backbone computes in)."""

    def __init__(self, batch, shape, cap, device, layout=LAYOUT_LINEAR, zero_count=True):
        lib = L.load()
        self.batch = int(batch)
        self.shape = [int(s) for s in shape]           # (D, H, W)
        self.cap = int(cap)
        self.layout = int(layout)
        nwords = lib.dz_index_words(self.batch, *self.shape, self.layout)
        self.bitmap = torch.empty((nwords,), dtype=torch.int32, device=device)
        self.prefix = torch.empty((nwords,), dtype=torch.int32, device=device)
        self.coords = torch.empty((max(self.cap, 1), 4), dtype=torch.int32, device=device)
        # (zero_count=False: the caller builds the level right away - every build ends with the scan writing the count)
        self.d_m = (torch.zeros if zero_count else torch.empty)((1,), dtype=torch.int32, device=device)
        self.ws = _ws(lib.dz_index_workspace_bytes(self.batch, *self.shape, self.layout))
        self._m_host = None
        # True for levels written by dz_voxelize_to_level: prefix[] is valid only at words that hold a bit, so only ACTIVE cells may
        # be ranked (csrc/common.h: bitmap_rank contract); build_from_coords / downsample write the full prefix
        self.prefix_partial = False
        # power-of-two pre-scale of the feature rows stored on this level (det_modules._Cached.set_prescale): rows hold value * 2^act_exp
        self.act_exp = 0

    def num_active(self):
        """Host copy of the active-site count (one sync; cached)."""
        if self._m_host is None:
            self._m_host = int(self.d_m.item())
        return self._m_host

    def build_from_coords(self, coords, d_n=None, want_rank=True):
        """coords (n,4) i32 [b,z,y,x] any order -> fills the level, returns rank_of_input (n,) i32."""
        lib = L.load()
        L.require_cuda(coords)
        n = coords.shape[0]
        rank = torch.empty((max(n, 1),), dtype=torch.int32, device=coords.device) if want_rank else None
        rc = lib.dz_index_from_coords(L.ptr(coords), L.ptr(d_n), n, self.batch, *self.shape, self.layout, L.ptr(self.bitmap),
                                      L.ptr(self.prefix), L.ptr(self.coords), L.ptr(self.d_m), self.cap,
                                      L.ptr(rank), L.ptr(self.ws), self.ws.numel(), L.stream())
        L.check(rc, 'dz_index_from_coords')
        self._m_host = None
        self.prefix_partial = False
        return rank

    def downsample(self, k, s, p, cap=None):
        """Output level of a strided sparse conv over this level."""
        lib = L.load()
        oshape = [(self.shape[i] + 2 * p[i] - k[i]) // s[i] + 1 for i in range(3)]
        cells = self.batch * oshape[0] * oshape[1] * oshape[2]
        if cap is None:
            per_in = 1
            for i in range(3):
                per_in *= (k[i] + s[i] - 1) // s[i]
            cap = min(cells, self.cap * per_in)
        out = SparseLevel(self.batch, oshape, cap, self.coords.device, layout=self.layout, zero_count=False)
        rc = lib.dz_index_downsample(L.ptr(self.coords), L.ptr(self.d_m), self.cap, self.batch, *self.shape, self.layout,
                                     L.i3(k), L.i3(s), L.i3(p), L.ptr(out.bitmap), L.ptr(out.prefix),
                                     L.ptr(out.coords), L.ptr(out.d_m), out.cap, L.ptr(out.ws), out.ws.numel(),
                                     L.stream())
        L.check(rc, 'dz_index_downsample')
        return out

    def neighbors_to(self, out_level, k, s, p, packed=False):
        """(kvol, out_level.cap) i32 neighbour table: rows of THIS level feeding each output row.
        packed=True asks for the packed form of a 27-tap table (dz_build_neighbors_packed: (9, cap) words, `.packed` = True,
        `.kvol` = 27) - what the small-channel split-math convolutions read; windows / layouts it does not cover come back
        unpacked."""
        lib = L.load()
        kvol = k[0] * k[1] * k[2]
        cap = max(out_level.cap, 1)
        masks = torch.empty((lib.dz_tile_masks_words(cap),), dtype=torch.int32, device=self.coords.device)
        args = (L.ptr(out_level.coords), L.ptr(out_level.d_m), out_level.cap, L.ptr(self.bitmap), L.ptr(self.prefix), self.batch,
                *self.shape, self.layout, L.i3(k), L.i3(s), L.i3(p))
        if packed and kvol == 27 and self.layout == LAYOUT_LINEAR and p[2] == 1 and cap < (1 << 29):
            nbr = torch.empty((9, cap), dtype=torch.int32, device=self.coords.device)
            rc = lib.dz_build_neighbors_packed(*args, L.ptr(nbr), L.ptr(masks), L.stream())
            if rc == 0:
                nbr.tile_masks = masks
                nbr.packed, nbr.kvol = True, 27
                return nbr
            if rc != L.ERR_UNSUPPORTED:
                L.check(rc, 'dz_build_neighbors_packed')
        nbr = torch.empty((kvol, cap), dtype=torch.int32, device=self.coords.device)
        rc = lib.dz_build_neighbors(*args, L.ptr(nbr), L.ptr(masks), L.stream())
        L.check(rc, 'dz_build_neighbors')
        nbr.tile_masks = masks        # per-32-row tap masks ride along with the table (consumed by spconv_forward)
        return nbr


FUSED_XWIN = os.environ.get('DZ_TUNE_FUSED_XWIN', '1') != '0'      # development switch: 0 = table and windows by two launches (r04)


def neighbors_xrun(level, channels):
    """Packed submanifold 3 x 3 x 3 table of `level` onto itself WITH the x-run windows (and, from XRUN_SORT_MIN_CHANNELS up, the
    tap-set order) from one launch (dz_build_neighbors_packed_x) - what `build_windows(level.neighbors_to(level, ..., packed=True),
    level, channels)` returns, bit for bit.  Levels / widths the engine does not cover come back as the plain packed / unpacked table."""
    lib = L.load()
    k3, s1, p1 = (3, 3, 3), (1, 1, 1), (1, 1, 1)
    tr = lib.dz_spconv_x_tile_rows(int(channels), int(channels))
    cap = max(level.cap, 1)
    if not FUSED_XWIN or tr == 0 or level.layout != LAYOUT_LINEAR or cap >= (1 << 29):
        return build_windows(level.neighbors_to(level, k3, s1, p1, packed=True), level, channels)
    dev = level.coords.device
    masks = torch.empty((lib.dz_tile_masks_words(cap),), dtype=torch.int32, device=dev)
    nbr = torch.empty((9, cap), dtype=torch.int32, device=dev)
    win = torch.empty((lib.dz_spconv_x_windows_words(cap, tr),), dtype=torch.int32, device=dev)
    sort = XRUN_SORT and channels >= XRUN_SORT_MIN_CHANNELS
    nbr_sorted = torch.empty_like(nbr) if sort else None
    perm = torch.empty((cap,), dtype=torch.int32, device=dev) if sort else None
    rc = lib.dz_build_neighbors_packed_x(L.ptr(level.coords), L.ptr(level.d_m), level.cap, L.ptr(level.bitmap), L.ptr(level.prefix), level.batch,
                                         *level.shape, L.ptr(nbr), L.ptr(masks), tr, L.ptr(win), L.ptr(nbr_sorted), L.ptr(perm), L.stream())
    L.check(rc, 'dz_build_neighbors_packed_x')
    nbr.tile_masks = masks
    nbr.packed, nbr.kvol = True, 27
    nbr.xwin = (win, tr, nbr_sorted, perm)
    return nbr


def unpack_table(nbr):
    """The (27, cap) form of a packed neighbour table (tests, statistics); other tables are returned as they are."""
    if not getattr(nbr, 'packed', False):
        return nbr
    e = nbr.to(torch.int64) & 0xFFFFFFFF
    r = e & 0x1FFFFFFF
    left, cen, right = (e >> 29) & 1, (e >> 30) & 1, (e >> 31) & 1
    minus = torch.full_like(r, -1)
    taps = torch.stack([torch.where(left == 1, r - 1, minus), torch.where(cen == 1, r, minus), torch.where(right == 1, r + cen, minus)], dim=1)
    return taps.reshape(27, nbr.shape[1]).to(torch.int32)


def table_pairs(nbr, m):
    """Number of (input, output) pairs of the first m output rows of a neighbour table (host sync)."""
    if getattr(nbr, 'packed', False):
        e = nbr[:, :m].to(torch.int64) & 0xFFFFFFFF
        return int((((e >> 29) & 1) + ((e >> 30) & 1) + ((e >> 31) & 1)).sum().item())
    return int((nbr[:, :m] >= 0).sum().item())


def build_tiles(nbr, out_level):
    """Tile form of a neighbour table for the tile-resident convolution (dz_spconv_tiles_forward): per tile of
    dz_spconv_tile_rows() output rows the list of distinct input rows (halo), the tile record (slots = non-empty taps, halo
    size, fragment slot masks), the local uint16 table and the tap-set order of the tile's rows (include/detzero_hip.h).
    Attached to the table as ``nbr.tiles`` = (halo, tinfo, ltab, rowmap) - spconv_forward picks the tile kernel whenever it is
    there (split math modes)."""
    lib = L.load()
    kvol, cap = nbr.shape
    tr = lib.dz_spconv_tile_rows()
    ntiles = (cap + tr - 1) // tr
    dev = nbr.device
    halo = torch.empty((ntiles, lib.dz_build_tiles_halo_stride(kvol)), dtype=torch.int32, device=dev)
    tinfo = torch.zeros((ntiles, lib.dz_spconv_tile_info_words()), dtype=torch.int32, device=dev)
    ltab = torch.empty((ntiles, lib.dz_spconv_tile_table_entries()), dtype=torch.int16, device=dev)
    rowmap = torch.empty((ntiles, tr), dtype=torch.int16, device=dev)
    rc = lib.dz_build_tiles(L.ptr(nbr), kvol, cap, L.ptr(out_level.d_m), L.ptr(halo), L.ptr(tinfo), L.ptr(ltab), L.ptr(rowmap), L.stream())
    L.check(rc, 'dz_build_tiles')
    nbr.tiles = (halo, tinfo, ltab, rowmap)
    return nbr


_XWIN_MISMATCH_LOGGED = False
XRUN_SORT = os.environ.get('DZ_TUNE_XRUN_SORT', '1') != '0'       # development switch: rows of a unit in tap-set order
XRUN_SORT_MIN_CHANNELS = int(os.environ.get('DZ_TUNE_XRUN_SORT_MIN', '64'))


def build_windows(nbr, out_level, channels):
    """Windows of the x-run convolution (dz_spconv_forward_split_x) for a PACKED submanifold table and the level's channel width:
    per tile of dz_spconv_x_tile_rows(channels, channels) output rows and z offset the contiguous range of input rows its nine taps
    read.  Attached as ``nbr.xwin`` = (windows, tile_rows); spconv_forward picks the x-run kernel whenever it is there and the
    layer is channels -> channels.  Tables / widths the engine does not cover are returned unchanged."""
    lib = L.load()
    tr = lib.dz_spconv_x_tile_rows(int(channels), int(channels))
    if not getattr(nbr, 'packed', False) or tr == 0:
        return nbr
    cap = nbr.shape[1]
    # (tiles x 3 x 2 window words + the tile-queue words of the convolution kernels)
    win = torch.empty((lib.dz_spconv_x_windows_words(cap, tr),), dtype=torch.int32, device=nbr.device)
    # the table and the row map in tap-set order (rows of a unit sorted by their neighbour pattern: fewer (fragment, tap) pairs to multiply)
    # (at 32 channels the order saves ~1 % of the kernel - less than sorting a level of 1.8 M rows and writing its table again costs)
    sort = XRUN_SORT and channels >= XRUN_SORT_MIN_CHANNELS
    nbr_sorted = torch.empty_like(nbr) if sort else None
    perm = torch.empty((cap,), dtype=torch.int32, device=nbr.device) if sort else None
    rc = lib.dz_spconv_x_windows(L.ptr(nbr), cap, L.ptr(out_level.d_m), tr, L.ptr(win), L.ptr(nbr_sorted), L.ptr(perm), L.stream())
    L.check(rc, 'dz_spconv_x_windows')
    nbr.xwin = (win, tr, nbr_sorted, perm)
    return nbr


def scatter_rows(src, rank, c_dst, cap, d_n=None, math=0):
    """dst[rank[i]] = src[i] (zero padded to c_dst channels); with math != 0 the rows are written as pair16."""
    lib = L.load()
    L.require_cuda(src, rank)
    n, c_src = src.shape
    dst = torch.zeros((max(cap, 1), c_dst), dtype=torch.float32, device=src.device)
    if math:
        rc = lib.dz_scatter_rows_split(L.ptr(src), L.ptr(rank), L.ptr(d_n), n, c_src, L.ptr(dst), c_dst, storage_math(math), L.stream())
    else:
        rc = lib.dz_scatter_rows(L.ptr(src), L.ptr(rank), L.ptr(d_n), n, c_src, L.ptr(dst), c_dst, L.stream())
    L.check(rc, 'dz_scatter_rows')
    return dst


def gather_rows(src, idx, d_n, n_cap):
    lib = L.load()
    c = src.shape[1]
    out = torch.zeros((max(n_cap, 1), c), dtype=torch.float32, device=src.device)
    rc = lib.dz_gather_rows(L.ptr(src), L.ptr(idx), L.ptr(d_n), n_cap, c, L.ptr(out), L.stream())
    L.check(rc, 'dz_gather_rows')
    return out


def spconv_forward(feats, nbr, out_level, w_taps, scale, shift, residual=None, relu=True, out=None, in_level=None, math=0, cout=None):
    """feats (m_in,cin); nbr (kvol,cap); returns (cap,cout).
    math == 0: fp32 rows, w_taps (kvol,cin,cout) fp32.
    math != 0: pair16 rows (in, residual, out), w_taps (kvol,cout_pad,cin) pair16 from pack_weight_split."""
    lib = L.load()
    L.require_cuda(feats, nbr, w_taps, scale, shift, residual)
    kvol, cap = nbr.shape
    packed = getattr(nbr, 'packed', False)
    if packed:
        kvol = nbr.kvol
        if not math:
            raise L.DetZeroHipError('spconv_forward: a packed neighbour table feeds the split-math kernels only')
    if math:
        # (split weights are padded to 32 output channels: the true count comes from the BatchNorm vector, or `cout=`)
        cin, cout = w_taps.shape[2], (int(cout) if cout is not None else scale.shape[0] if scale is not None else w_taps.shape[1])
    else:
        cin, cout = w_taps.shape[1], w_taps.shape[2]
    assert feats.shape[1] == cin, (feats.shape, w_taps.shape)
    if out is None:
        out = torch.empty((cap, cout), dtype=torch.float32, device=feats.device)

    tiles = getattr(nbr, 'tiles', None) if (math and kvol >= 3) else None
    xwin = getattr(nbr, 'xwin', None) if (math and packed and cin == cout) else None
    if xwin is not None and lib.dz_spconv_x_tile_rows(cin, cout) != xwin[1]:
        # (windows built for another tile size - e.g. DZ_TUNE_X32 changed after the index was built: the packed gather kernel runs instead)
        global _XWIN_MISMATCH_LOGGED
        if not _XWIN_MISMATCH_LOGGED:
            _XWIN_MISMATCH_LOGGED = True
            import warnings
            warnings.warn('spconv_forward: windows built for %d-row units, the %d-channel x-run kernel uses %d - falling back to the packed '
                          'gather kernel for this table' % (xwin[1], cout, lib.dz_spconv_x_tile_rows(cin, cout)))
        xwin = None

    def launch():
        if xwin is not None:
            rc = lib.dz_spconv_forward_split_x(L.ptr(feats), feats.shape[0], cin, L.ptr(xwin[2] if xwin[3] is not None else nbr), L.ptr(xwin[3]),
                                               L.ptr(xwin[0]), xwin[1], cap,
                                               L.ptr(out_level.d_m), L.ptr(w_taps), L.ptr(scale), L.ptr(shift), L.ptr(residual),
                                               1 if relu else 0, L.ptr(out), cout, int(math), L.stream())
        elif tiles is not None:
            rc = lib.dz_spconv_tiles_forward(L.ptr(feats), feats.shape[0], cin, L.ptr(tiles[0]), L.ptr(tiles[1]), L.ptr(tiles[2]),
                                             L.ptr(tiles[3]), kvol, cap, L.ptr(out_level.d_m), L.ptr(w_taps), L.ptr(scale),
                                             L.ptr(shift), L.ptr(residual), 1 if relu else 0, L.ptr(out), cout, int(math), L.stream())
        elif packed:
            rc = lib.dz_spconv_forward_split_packed(L.ptr(feats), feats.shape[0], cin, L.ptr(nbr), L.ptr(nbr.tile_masks), cap,
                                                    L.ptr(out_level.d_m), L.ptr(w_taps), L.ptr(scale), L.ptr(shift), L.ptr(residual),
                                                    1 if relu else 0, L.ptr(out), cout, int(math), L.stream())
        elif math:
            rc = lib.dz_spconv_forward_split(L.ptr(feats), feats.shape[0], cin, L.ptr(nbr), L.ptr(getattr(nbr, 'tile_masks', None)),
                                             kvol, cap, L.ptr(out_level.d_m),
                                             L.ptr(w_taps), L.ptr(scale), L.ptr(shift), L.ptr(residual), 1 if relu else 0,
                                             L.ptr(out), cout, int(math), L.stream())
        else:
            rc = lib.dz_spconv_forward(L.ptr(feats), feats.shape[0], cin, L.ptr(nbr), kvol, cap, L.ptr(out_level.d_m),
                                       L.ptr(w_taps), L.ptr(scale), L.ptr(shift), L.ptr(residual), 1 if relu else 0,
                                       L.ptr(out), cout, L.stream())
        L.check(rc, 'dz_spconv_forward')
    if PROFILER is None:
        launch()
    else:
        # algorithmic work of this launch: 2*pairs*cin*cout FLOP; bytes per BASELINE.md section 3
        m = out_level.num_active()
        pairs = table_pairs(nbr, m)
        flops = 2.0 * pairs * cin * cout
        n_in = in_level.num_active() if in_level is not None else m
        nbytes = 4.0 * (n_in * cin + m * cout + kvol * cin * cout + (m * cout if residual is not None else 0)) + 8.0 * pairs
        name = (lib.dz_spconv_x_variant(cin, cout) if xwin is not None else lib.dz_spconv_tiles_variant(cin, cout) if tiles is not None else
                lib.dz_spconv_variant_split(cin, cout) if math else lib.dz_spconv_variant(cin, cout))
        PROFILER.wrap(name.decode(), flops, nbytes, launch)
    return out


def bev_row_index(level, feat_rows, pad=1):
    """(B, H + 2 pad, W + 2 pad, 2) int32: the feature row of every (pixel, z slab) of a two-slab level, -1 = empty / border / past
    `feat_rows` - the sparse-input form of HeightCompression (dz_conv2d_desc.in_rowidx)."""
    lib = L.load()
    d, h, w = level.shape
    idx = torch.empty((level.batch, h + 2 * pad, w + 2 * pad, 2), dtype=torch.int32, device=level.coords.device)
    rc = lib.dz_bev_row_index(L.ptr(level.bitmap), L.ptr(level.prefix), level.batch, d, h, w, level.layout, pad, int(feat_rows), L.ptr(idx), L.stream())
    L.check(rc, 'dz_bev_row_index')
    return idx


def bev_tile_list(ridx, ho, wo, nlists=1):
    """(nlists, words) int32: list l - 1 = [n to run, n skippable, tiles to run ascending, skippable ones] for the l-th 3 x 3 layer of the
    first BEV block over a row-index image (dz_bev_tile_list: 8 x 32 pixel tiles whose pixels are all at least l + 1 pixels away from any row)."""
    lib = L.load()
    b, hp, wp, _ = ridx.shape
    n = lib.dz_bev_tile_list_words(b, int(ho), int(wo))
    lst = torch.empty((int(nlists), n), dtype=torch.int32, device=ridx.device)
    ws = torch.empty((n,), dtype=torch.uint8, device=ridx.device)
    L.check(lib.dz_bev_tile_list(L.ptr(ridx), b, hp, wp, int(ho), int(wo), int(nlists), L.ptr(lst), L.ptr(ws), L.stream()), 'dz_bev_tile_list')
    return lst


def bev_fill_empty_tiles(tiles, batch, ho, wo, shift, relu, cout, out, math, zero_resp=None):
    """The skippable pixel tiles of list `tiles` in the zero-bordered pair16 image `out` (B, ho + 2, wo + 2, C) get the layer's zero-input
    response: the (1, ho + 2, wo + 2, C) image `zero_resp`, or - None: the first layer - the constant ReLU(shift)."""
    lib = L.load()
    L.check(lib.dz_bev_fill_empty_tiles(L.ptr(tiles), int(batch), int(ho), int(wo), L.ptr(shift), 1 if relu else 0, int(cout), L.ptr(out), out.shape[1],
                                        out.shape[2], out.shape[3], 0, L.ptr(zero_resp), int(math), L.stream()), 'dz_bev_fill_empty_tiles')


BEV_DENSE = not os.environ.get('DZ_BEV_SCATTER')   # development switch: False = zero-fill + scatter (dz_sparse_to_bev_split) for two-slab levels too


def sparse_to_bev(feats, level, c, pad=1, out=None, math=0):
    """-> (B, H+2p, W+2p, C*D) channel-last zero-bordered BEV image (pair16 in, pair16 out when math != 0)."""
    lib = L.load()
    d, h, w = level.shape
    if out is None:
        out = torch.empty((level.batch, h + 2 * pad, w + 2 * pad, c * d), dtype=torch.float32, device=feats.device)
    if math and d == 2 and c % 8 == 0 and BEV_DENSE:
        # two z slabs (the backbone's encoded tensor): the whole image, zeros included, written once from the level's own index
        rc = lib.dz_sparse_to_bev_split_dense(L.ptr(feats), feats.shape[0], L.ptr(level.bitmap), L.ptr(level.prefix), level.batch, c, d, h, w, level.layout,
                                              pad, L.ptr(out), L.stream())
        L.check(rc, 'dz_sparse_to_bev_split_dense')
        return out
    out.zero_()
    fn = lib.dz_sparse_to_bev_split if math else lib.dz_sparse_to_bev
    rc = fn(L.ptr(feats), L.ptr(level.coords), L.ptr(level.d_m), level.cap, c, d, h, w, pad, L.ptr(out), L.stream())
    L.check(rc, 'dz_sparse_to_bev')
    return out


# ------------------------------------------------------------------------------------------------
# split precision (pair16): csrc/hgemm.h
# ------------------------------------------------------------------------------------------------
# 'f16' (3): the tensors of 'f16x2' (fp16 pairs), but every convolution product is ONE fp16 MFMA on the hi halves - plain-fp16
# inputs with fp32 accumulation, a third of the matrix work.  Opt-in fast mode, NOT fp32-class.
MATH_MODES = {'f32': 0, 'f16x2': 1, 'bf16x2': 2, 'f16': 3}


def storage_math(math):
    """The pair16 encoding a math mode keeps its tensors in (3 = 'f16' computes on the tensors of 1 = 'f16x2')."""
    return 1 if int(math) == 3 else int(math)


def math_id(mode):
    if isinstance(mode, str):
        if mode not in MATH_MODES:
            raise L.DetZeroHipError('unknown math mode %r (choose from %s)' % (mode, sorted(MATH_MODES)))
        return MATH_MODES[mode]
    return int(mode)


def pair16_pack(x, math):
    """Host/torch packer (any device): x (..., C) fp32, C % 8 == 0 -> same-shape float32-typed pair16 bits.
    Used for weights at plan time; the device kernels (dz_pair16_from_f32, conv epilogues) do the same split."""
    math = storage_math(math)
    dt = torch.float16 if math == 1 else torch.bfloat16
    x = x.float()
    if math == 1:
        x = x.clamp(-65504.0, 65504.0)
    hi = x.to(dt)
    lo = (x - hi.float()).to(dt)
    shp = x.shape
    g = shp[-1] // 8
    out = torch.stack([hi.reshape(*shp[:-1], g, 8), lo.reshape(*shp[:-1], g, 8)], dim=-2)
    return out.reshape(*shp[:-1], g * 16).contiguous().view(torch.float32)


def pair16_unpack(x, math):
    math = storage_math(math)
    dt = torch.float16 if math == 1 else torch.bfloat16
    shp = x.shape
    g = shp[-1] // 8
    h = x.contiguous().view(dt).reshape(*shp[:-1], g, 2, 8).float()
    return (h[..., 0, :] + h[..., 1, :]).reshape(shp)


def pack_weight_split(w, math, cout_mult=32):
    """(..., cin, cout) fp32 conv weights -> (..., cout_pad, cin) pair16 (cout padded with zero rows)."""
    wt = w.float().transpose(-1, -2)
    co = wt.shape[-2]
    cp = -(-co // cout_mult) * cout_mult
    if cp > co:
        wt = torch.cat([wt, wt.new_zeros(*wt.shape[:-2], cp - co, wt.shape[-1])], dim=-2)
    return pair16_pack(wt.contiguous(), math)


F16_PAIR_WEIGHT_TOP = 2.0 ** 14      # where the largest |weight| of an output channel is placed before the fp16-pair split


def weight_prescale(w):
    """Exact per-output-channel power-of-two scaling of conv / linear weights (..., cin, cout) for fp16-pair storage.
    An fp16 pair carries 22 significant bits only while its lo half is a NORMAL fp16 (|value| >= 2^-3); checkpoint weights are
    O(1e-2) and would keep ~18.  Every output channel is multiplied by 2^e_c so that its largest |weight| lands in [2^13, 2^15)
    (fp16 tops out at 65504; the products are accumulated in fp32, where a power of two changes nothing), and 2^-e_c goes into the
    layer's per-channel epilogue scale.  Returns (scaled weights, inverse factors (..., cout))."""
    red = (-3, -2) if w.dim() >= 3 else (-2,)
    amax = w.detach().abs().float().amax(dim=red)
    e = torch.floor(torch.log2(F16_PAIR_WEIGHT_TOP / amax.clamp_min(1e-30))).clamp_(-40.0, 40.0)
    e = torch.where(amax > 0, e, torch.zeros_like(e)).to(torch.int32)
    one = torch.ones_like(amax)
    up, down = torch.ldexp(one, e), torch.ldexp(one, -e)
    for _ in red:
        up = up.unsqueeze(-2)
    return w.float() * up, down


def level_rows_f32(rows, level, math=0):
    """Feature rows of a level as plain fp32 values: pair16 decoded, the level's power-of-two pre-scale (SparseLevel.act_exp) removed."""
    plain = pair16_to_f32(rows, math) if math else rows
    e = int(getattr(level, 'act_exp', 0) or 0)
    return plain * (2.0 ** -e) if e else plain


def pair16_from_f32(x, c_dst=None, math=1):
    """device conversion of rows (..., C) fp32 -> (..., c_dst) pair16"""
    lib = L.load()
    L.require_cuda(x)
    x = x.contiguous()
    c = x.shape[-1]
    c_dst = c if c_dst is None else c_dst
    rows = x.numel() // c
    out = torch.empty((*x.shape[:-1], c_dst), dtype=torch.float32, device=x.device)
    L.check(lib.dz_pair16_from_f32(L.ptr(x), rows, c, c_dst, storage_math(math), L.ptr(out), L.stream()), 'dz_pair16_from_f32')
    return out


def pair16_to_f32(x, math=1):
    lib = L.load()
    L.require_cuda(x)
    x = x.contiguous()
    c = x.shape[-1]
    out = torch.empty_like(x)
    L.check(lib.dz_pair16_to_f32(L.ptr(x), x.numel() // c, c, storage_math(math), L.ptr(out), L.stream()), 'dz_pair16_to_f32')
    return out


# ------------------------------------------------------------------------------------------------
# dense conv
# ------------------------------------------------------------------------------------------------
def conv2d(desc_kwargs, math=0, out_f32=False, tiles=None):
    """One dense-conv launch from a dict of dz_conv2d_desc fields.  math != 0: pair16 input / weights
    (pack_weight_split layout) and pair16 output unless out_f32.  tiles: the tensor behind `in_tiles` (profiling only: the work of
    the launch is that of the tiles it runs)."""
    lib = L.load()
    d = L.Conv2dDesc()
    g_cout = desc_kwargs.pop('g_cout')
    g_ooff = desc_kwargs.pop('g_ooff')
    for k, v in desc_kwargs.items():
        setattr(d, k, v)
    for i, v in enumerate(g_cout):
        d.g_cout[i] = int(v)
    for i, v in enumerate(g_ooff):
        d.g_ooff[i] = int(v)

    def launch():
        if math:
            L.check(lib.dz_conv2d_forward_split(ctypes.byref(d), int(math), 1 if out_f32 else 0, L.stream()), 'dz_conv2d_forward_split')
        else:
            L.check(lib.dz_conv2d_forward(ctypes.byref(d), L.stream()), 'dz_conv2d_forward')
    if PROFILER is None:
        launch()
        return
    m = d.batch * d.ho * d.wo
    taps = d.kh * d.kw
    cout = sum(d.g_cout[i] for i in range(d.groups))
    flops = 2.0 * m * taps * d.cin * cout
    nbytes = 4.0 * (m * d.cin * (1 if d.phase_groups else d.groups) + m * cout + taps * d.cin * d.cout_pad * d.groups)
    if tiles is not None:
        # zero-response tiles are not run: count the EXECUTED share of the layer (one host sync; profiling passes only), so that the
        # roofline figures of the kernel describe the kernel, not the work it was spared
        n_run, n_skip = (int(v) for v in tiles[:2].tolist())
        share = n_run / float(max(n_run + n_skip, 1))
        flops *= share
        nbytes = 4.0 * (share * (m * d.cin + m * cout) + taps * d.cin * d.cout_pad * d.groups)
    name = (lib.dz_conv2d_variant_split if math else lib.dz_conv2d_variant)(ctypes.byref(d)).decode()
    PROFILER.wrap(name, flops, nbytes, launch)


def linear(x, w, scale, shift, relu, cout, out=None, group_shift=None, group_rows=0):
    """x (rows, cin) @ w (cin, cout_pad) -> (rows, cout); optional per-row-group pre-activation addend."""
    lib = L.load()
    L.require_cuda(x, w, scale, shift, group_shift)
    rows, cin = x.shape
    cout_pad = w.shape[1]
    if out is None:
        out = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
    rc = lib.dz_linear_forward(L.ptr(x), rows, cin, x.stride(0), L.ptr(w), cout, cout_pad, L.ptr(scale),
                               L.ptr(shift), L.ptr(group_shift), int(group_rows), 1 if relu else 0, L.ptr(out),
                               out.stride(0), L.stream())
    L.check(rc, 'dz_linear_forward')
    return out


SPLITK_MAX_ROWS, SPLITK_MIN_CIN = 2048, 4096


def linear_splitk_ok(rows, cin):
    return 0 < rows <= SPLITK_MAX_ROWS and cin >= SPLITK_MIN_CIN and cin % 256 == 0


def linear_splitk(x, w, scale, shift, relu, cout, splits=8):
    """dz_linear_forward_splitk: ops.linear for a few rows against a very long input (cin % (splits * 32) == 0): the input channels in
    `splits` groups side by side, summed in a second pass."""
    lib = L.load()
    L.require_cuda(x, w, scale, shift)
    rows, cin = x.shape
    cout_pad = w.shape[1]
    nbytes = lib.dz_linear_splitk_workspace_bytes(rows, cout_pad, splits)
    ws = torch.empty(((nbytes + 3) // 4,), dtype=torch.float32, device=x.device)
    out = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
    rc = lib.dz_linear_forward_splitk(L.ptr(x), rows, cin, x.stride(0), L.ptr(w), cout, cout_pad, L.ptr(scale), L.ptr(shift), 1 if relu else 0, L.ptr(out),
                                      out.stride(0), int(splits), L.ptr(ws), nbytes, L.stream())
    L.check(rc, 'dz_linear_forward_splitk')
    return out


def linear_split(x, w, scale, shift, relu, cout, math, out_f32=False, group_shift=None, group_rows=0, group_max=False):
    """dz_linear_forward_split: x (rows, >= cin words) pair16 @ w (cout_pad, cin) pair16 -> (rows, cout) fp32 when out_f32,
    else (rows, cout) pair16 (cout % 32 == 0 so that the result can feed the next layer).
    group_max: -> (rows / group_rows, cout) fp32, the maximum over each group of `group_rows` consecutive rows, taken in the layer's
    epilogue (the PointNet max over an object's points; the (rows, cout) activation is never written)."""
    lib = L.load()
    L.require_cuda(x, w, scale, shift, group_shift)
    rows = x.shape[0]
    cout_pad, cin = w.shape
    if not out_f32 and cout % 32:
        raise L.DetZeroHipError('linear_split: a pair16 result needs cout %% 32 == 0 (got %d)' % cout)
    if group_shift is not None and group_shift.shape[1] != cout_pad:
        raise L.DetZeroHipError('linear_split: group_shift rows must be cout_pad = %d wide' % cout_pad)
    if group_max and (group_rows < 128 or group_rows % 128 or rows % group_rows):
        raise L.DetZeroHipError('linear_split: group_max needs groups of a multiple of 128 rows (got %d rows in groups of %d)' % (rows, group_rows))
    out = torch.empty((rows // group_rows if group_max else rows, cout), dtype=torch.float32, device=x.device)
    rc = lib.dz_linear_forward_split(L.ptr(x), rows, cin, x.stride(0), L.ptr(w), cout, cout_pad, L.ptr(scale), L.ptr(shift),
                                     L.ptr(group_shift), int(group_rows), 1 if relu else 0, L.ptr(out), out.stride(0), int(math),
                                     1 if (out_f32 or group_max) else 0, 1 if group_max else 0, L.stream())
    L.check(rc, 'dz_linear_forward_split')
    return out


def pointnet3(x, layers, group_rows, math, want_tap=False, x_f32=False):
    """dz_pointnet3_forward: x (rows, 32) pair16 - or, with x_f32, (rows, 16 | 32) fp32 rows split inside the kernel - through three
    (w (cout_pad, cin) pair16, scale, shift) layers 32 -> 128 -> 128 -> c3 with ReLU and the max over every `group_rows` rows ->
    (pooled (rows / group_rows, c3) fp32, tap (rows, 128) pair16 or None)."""
    lib = L.load()
    (w1, s1, b1), (w2, s2, b2), (w3, s3, b3) = layers
    L.require_cuda(x, w1, w2, w3, s1, b1, s2, b2, s3, b3)
    rows = x.shape[0]
    c3 = w3.shape[0]
    if tuple(w1.shape) != (128, 32) or tuple(w2.shape) != (128, 128) or w3.shape[1] != 128 or x.shape[1] not in ((16, 32) if x_f32 else (32,)) or rows % group_rows:
        raise L.DetZeroHipError('pointnet3: expects 32 -> 128 -> 128 -> c3 layers on (rows, 32) pair16 input (got %s, %s, %s on %s)' % (
            tuple(w1.shape), tuple(w2.shape), tuple(w3.shape), tuple(x.shape)))
    out = torch.empty((rows // group_rows, c3), dtype=torch.float32, device=x.device)
    tap = torch.empty((rows, 128), dtype=torch.float32, device=x.device) if want_tap else None
    rc = lib.dz_pointnet3_forward(L.ptr(x), rows, L.ptr(w1), L.ptr(s1), L.ptr(b1), L.ptr(w2), L.ptr(s2), L.ptr(b2), L.ptr(w3), L.ptr(s3), L.ptr(b3),
                                  c3, int(group_rows), L.ptr(tap), L.ptr(out), x.shape[1] if x_f32 else 0, storage_math(math), L.stream())
    L.check(rc, 'dz_pointnet3_forward')
    return out, tap


def mlp_chain(x, la, lb, group_shift, group_rows, math, kv=None):
    """dz_mlp_chain_forward: x (rows, 128) pair16 through la = (w (512, 128) pair16, scale, shift) with the per-group pre-BatchNorm addend
    group_shift (groups, >= 512) and lb = (w (256, 512) pair16, scale, shift), both with ReLU -> memory (rows, 256) fp32; with
    kv = (wk (256, 256) pair16, bk, wv, bv) also the key / value projections of the memory -> (memory, k, v)."""
    lib = L.load()
    (wa, sa, ba), (wb, sb, bb) = la, lb
    L.require_cuda(x, wa, sa, ba, wb, sb, bb)
    rows = x.shape[0]
    if tuple(wa.shape) != (512, 128) or tuple(wb.shape) != (256, 512) or x.shape[1] != 128:
        raise L.DetZeroHipError('mlp_chain: expects 128 -> 512 -> 256 layers on (rows, 128) pair16 rows (got %s, %s on %s)' % (
            tuple(wa.shape), tuple(wb.shape), tuple(x.shape)))
    mem = torch.empty((rows, 256), dtype=torch.float32, device=x.device)
    k = v = wk = bk = wv = bv = None
    if kv is not None:
        wk, bk, wv, bv = kv
        if tuple(wk.shape) != (256, 256) or tuple(wv.shape) != (256, 256):
            raise L.DetZeroHipError('mlp_chain: key / value projections must be 256 -> 256')
        k, v = torch.empty_like(mem), torch.empty_like(mem)
    ldg = 0 if group_shift is None else group_shift.stride(0)
    rc = lib.dz_mlp_chain_forward(L.ptr(x), rows, L.ptr(wa), L.ptr(sa), L.ptr(ba), L.ptr(group_shift), ldg, int(group_rows), L.ptr(wb), L.ptr(sb), L.ptr(bb),
                                  L.ptr(wk), L.ptr(bk), L.ptr(wv), L.ptr(bv), L.ptr(mem), L.ptr(k), L.ptr(v), storage_math(math), L.stream())
    L.check(rc, 'dz_mlp_chain_forward')
    return (mem, k, v) if kv is not None else mem


def group_max(x, groups, length):
    """x (groups*length, c) -> (groups, c) max over each group's rows."""
    lib = L.load()
    L.require_cuda(x)
    c = x.shape[1]
    out = torch.empty((groups, c), dtype=torch.float32, device=x.device)
    rc = lib.dz_group_max(L.ptr(x), groups, length, c, L.ptr(out), L.stream())
    L.check(rc, 'dz_group_max')
    return out


def add_layernorm(x, y, gamma, beta, eps=1e-5, norm=True):
    """LayerNorm(x + y) (or x + y when norm=False) over the last dimension."""
    lib = L.load()
    L.require_cuda(x, y, gamma, beta)
    c = x.shape[-1]
    rows = x.numel() // c
    out = torch.empty_like(x)
    rc = lib.dz_add_layernorm(L.ptr(x), L.ptr(y), L.ptr(gamma), L.ptr(beta), rows, c, float(eps), 1 if norm else 0,
                              L.ptr(out), L.stream())
    L.check(rc, 'dz_add_layernorm')
    return out


def rows_all_zero(tensors):
    """(rows,) bool: every int of the row is zero in all of up to four (rows, w) int32 tensors."""
    lib = L.load()
    L.require_cuda(*tensors)
    ts = [t.contiguous() for t in tensors]
    rows = ts[0].shape[0]
    out = torch.empty((rows,), dtype=torch.uint8, device=ts[0].device)
    ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    ws = (ctypes.c_int * len(ts))(*[int(t.shape[1]) for t in ts])
    L.check(lib.dz_rows_all_zero(ptrs, ws, len(ts), rows, L.ptr(out), L.stream()), 'dz_rows_all_zero')
    return out.bool()


def add_layernorm_combine(x, y, gamma, beta, eps, post, group_skip, group_rows):
    """post + (group_skip[row // group_rows] ? post : LayerNorm(x + y)) in one pass (c = 192)."""
    lib = L.load()
    L.require_cuda(x, y, gamma, beta, post, group_skip)
    c = x.shape[-1]
    out = torch.empty_like(x)
    rc = lib.dz_add_layernorm_combine(L.ptr(x), L.ptr(y), L.ptr(gamma), L.ptr(beta), x.numel() // c, c, float(eps), L.ptr(post), L.ptr(group_skip),
                                      int(group_rows), L.ptr(out), L.stream())
    L.check(rc, 'dz_add_layernorm_combine')
    return out


# ------------------------------------------------------------------------------------------------
# head post-processing
# ------------------------------------------------------------------------------------------------
def centerhead_decode(head, h, w, ncls, k, score_thresh, limit6, pc_range, voxel_size, stride, use_iou=True):
    """head (B, H*W, 12) -> boxes (B,K,7), scores (B,K), labels (B,K) i32, counts (B,) i32 (device)."""
    lib = L.load()
    L.require_cuda(head)
    b = head.shape[0]
    dev = head.device
    boxes = torch.zeros((b, k, 7), dtype=torch.float32, device=dev)
    scores = torch.zeros((b, k), dtype=torch.float32, device=dev)
    labels = torch.zeros((b, k), dtype=torch.int32, device=dev)
    counts = torch.zeros((b,), dtype=torch.int32, device=dev)
    ws = _ws(lib.dz_centerhead_decode_workspace_bytes(b, h * w, ncls, k))
    rc = lib.dz_centerhead_decode(L.ptr(head), b, h, w, ncls, k, float(score_thresh), L.f6(limit6),
                                  L.f6(pc_range), L.f3(voxel_size), int(stride), 1 if use_iou else 0,
                                  L.ptr(boxes), L.ptr(scores), L.ptr(labels), L.ptr(counts), L.ptr(ws),
                                  ws.numel(), L.stream())
    L.check(rc, 'dz_centerhead_decode')
    return boxes, scores, labels, counts


def nms_rotated_nosync(boxes_sorted, d_n, thresh, post_max):
    """boxes_sorted (n_cap,7) descending score; returns keep (n_cap,) i32, d_num_keep (1,) i32."""
    lib = L.load()
    L.require_cuda(boxes_sorted)
    n_cap = boxes_sorted.shape[0]
    keep = torch.zeros((max(n_cap, 1),), dtype=torch.int32, device=boxes_sorted.device)
    d_nk = torch.zeros((1,), dtype=torch.int32, device=boxes_sorted.device)
    ws = _ws(lib.dz_nms_workspace_bytes(n_cap))
    rc = lib.dz_nms_rotated(L.ptr(boxes_sorted), L.ptr(d_n), n_cap, float(thresh), int(post_max), L.ptr(keep),
                            L.ptr(d_nk), L.ptr(ws), ws.numel(), L.stream())
    L.check(rc, 'dz_nms_rotated')
    return keep, d_nk


def nms_rotated_batched_nosync(boxes_sorted, d_n, thresh, post_max):
    """boxes_sorted (B,n_cap,7) descending score per item, d_n (B,) -> keep (B,n_cap) i32, d_num_keep (B,) i32."""
    lib = L.load()
    L.require_cuda(boxes_sorted, d_n)
    b, n_cap = boxes_sorted.shape[0], boxes_sorted.shape[1]
    keep = torch.zeros((b, max(n_cap, 1)), dtype=torch.int32, device=boxes_sorted.device)
    d_nk = torch.zeros((b,), dtype=torch.int32, device=boxes_sorted.device)
    ws = _ws(b * lib.dz_nms_workspace_bytes(n_cap))
    rc = lib.dz_nms_rotated_batched(L.ptr(boxes_sorted), L.ptr(d_n), b, n_cap, float(thresh), int(post_max), L.ptr(keep),
                                    L.ptr(d_nk), L.ptr(ws), ws.numel(), L.stream())
    L.check(rc, 'dz_nms_rotated_batched')
    return keep, d_nk


def pack_detections(boxes, scores, labels, keep, d_nk, post_max):
    """(B,K,7),(B,K),(B,K) i32, keep (B,K) i32, d_nk (B,) -> (B,post_max,9) [box7|score|label+1], zero rows after."""
    lib = L.load()
    L.require_cuda(boxes, scores, labels, keep, d_nk)
    b, k = boxes.shape[0], boxes.shape[1]
    out = torch.empty((b, post_max, 9), dtype=torch.float32, device=boxes.device)
    rc = lib.dz_pack_detections(L.ptr(boxes), L.ptr(scores), L.ptr(labels), L.ptr(keep), L.ptr(d_nk), b, k, int(post_max),
                                L.ptr(out), L.stream())
    L.check(rc, 'dz_pack_detections')
    return out


def boxes_pairwise(a, b, iou):
    lib = L.load()
    L.require_cuda(a, b)
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    fn = lib.dz_boxes_iou_bev if iou else lib.dz_boxes_overlap_bev
    rc = fn(L.ptr(a), a.shape[0], L.ptr(b), b.shape[0], L.ptr(out), L.stream())
    L.check(rc, 'dz_boxes_%s_bev' % ('iou' if iou else 'overlap'))
    return out


def points_in_boxes_v2(points, boxes):
    """points (B,M,3), boxes (B,T,7) -> (B,T,M) i32."""
    lib = L.load()
    L.require_cuda(points, boxes)
    b, m, _ = points.shape
    t = boxes.shape[1]
    mask = torch.zeros((b, t, m), dtype=torch.int32, device=points.device)
    rc = lib.dz_points_in_boxes_v2(L.ptr(boxes), L.ptr(points), b, t, m, L.ptr(mask), L.stream())
    L.check(rc, 'dz_points_in_boxes_v2')
    return mask


def crop_points_in_boxes_nosync(xyz, boxes, payload, cap):
    """xyz (M,3) f32, boxes (T,7) f32, payload (M,W) any 4-byte-multiple dtype -> (out (cap,W) payload dtype, index (cap,) i32,
    offsets (T+1,) i32, d_total (1,) i32): the rows of the points inside box 0, then box 1, ... (ascending point index)."""
    lib = L.load()
    L.require_cuda(xyz, boxes, payload)
    m, t = xyz.shape[0], boxes.shape[0]
    words = payload.shape[1] * payload.element_size() // 4
    assert payload.shape[0] == m and payload.shape[1] * payload.element_size() % 4 == 0
    dev = xyz.device
    out = torch.empty((max(cap, 1), payload.shape[1]), dtype=payload.dtype, device=dev)
    index = torch.empty((max(cap, 1),), dtype=torch.int32, device=dev)
    offsets = torch.zeros((t + 1,), dtype=torch.int32, device=dev)
    d_total = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = _ws(lib.dz_crop_points_workspace_bytes(m, t, cap))
    rc = lib.dz_crop_points_in_boxes(L.ptr(xyz), m, L.ptr(boxes), t, L.ptr(payload), words, L.ptr(out), L.ptr(index), L.ptr(offsets),
                                     L.ptr(d_total), cap, L.ptr(ws), ws.numel(), L.stream())
    L.check(rc, 'dz_crop_points_in_boxes')
    return out, index, offsets, d_total


def mha_core(q, k, v, key_padding_mask, heads, scale, math=0):
    """q (B,Lq,E), k/v (B,Lk,E), mask (B,Lk) bool/uint8 or None -> (B,Lq,E).  math 0: exact fp32 (dz_mha_core); 1 / 2: operands as
    16-bit pairs on the 16-bit matrix cores (dz_mha_core_split)."""
    lib = L.load()
    L.require_cuda(q, k, v)
    b, lq, e = q.shape
    lk = k.shape[1]
    if e != heads * 32:
        raise L.DetZeroHipError('dz_mha_core supports head_dim 32 only (E=%d, heads=%d)' % (e, heads))
    m8 = None
    if key_padding_mask is not None:
        m8 = key_padding_mask.to(torch.uint8).contiguous()
    out = torch.empty_like(q)
    if math:
        rc = lib.dz_mha_core_split(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(m8), b, lq, lk, heads, float(scale), L.ptr(out), storage_math(math), L.stream())
        L.check(rc, 'dz_mha_core_split')
        return out
    rc = lib.dz_mha_core(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(m8), b, lq, lk, heads, float(scale), L.ptr(out),
                         L.stream())
    L.check(rc, 'dz_mha_core')
    return out


def roi_bev_features(boxes, bev_hwc, x_lo, y_lo, voxel_x, voxel_y, stride):
    """dz_roi_bev_features: boxes (n, 7) of one frame, bev_hwc (H, W, C) view with channel stride 1 -> (n, 5 * C) bilinear BEV features at
    the box centre and the four edge middles (center_head.py:408-432,461-486)."""
    lib = L.load()
    if not boxes.is_cuda or not bev_hwc.is_cuda or bev_hwc.stride(2) != 1:
        raise L.DetZeroHipError('roi_bev_features: device tensors, BEV map as an (H, W, C) view with contiguous channels')
    boxes = boxes.float().contiguous()
    n = boxes.shape[0]
    h, w, c = bev_hwc.shape
    out = torch.empty((n, 5 * c), dtype=torch.float32, device=boxes.device)
    rc = lib.dz_roi_bev_features(L.ptr(boxes), n, bev_hwc.data_ptr(), bev_hwc.stride(0), bev_hwc.stride(1), h, w, c, float(x_lo), float(y_lo), float(voxel_x),
                                 float(voxel_y), int(stride), L.ptr(out), L.stream())
    L.check(rc, 'dz_roi_bev_features')
    return out


def xattn_folded_supported(lq, e, heads):
    return bool(L.load().dz_xattn_folded_supported(int(lq), int(e), int(heads)))


def xattn_folded(q, mem, key_padding_mask, wk_oi, wv_io, bv, heads, scale):
    """dz_xattn_folded: q (B,Lq,E) projected queries, mem (B,Lk,E) RAW memory rows, wk_oi = Wk (out, in), wv_io = Wv^T (in, out), bv (E)
    -> (B,Lq,E) attention output before out_proj; the key / value projections of the memory are never formed."""
    lib = L.load()
    L.require_cuda(q, mem, wk_oi, wv_io, bv)
    b, lq, e = q.shape
    lk = mem.shape[1]
    m8 = None if key_padding_mask is None else key_padding_mask.to(torch.uint8).contiguous()
    nbytes = lib.dz_xattn_folded_workspace_bytes(b, lk)
    ws = torch.empty(((nbytes + 3) // 4,), dtype=torch.float32, device=q.device)
    out = torch.empty_like(q)
    rc = lib.dz_xattn_folded(L.ptr(q), L.ptr(mem), L.ptr(m8), L.ptr(wk_oi), L.ptr(wv_io), L.ptr(bv), b, lq, lk, e, int(heads), float(scale),
                             L.ptr(ws), nbytes, L.ptr(out), L.stream())
    L.check(rc, 'dz_xattn_folded')
    return out


# ------------------------------------------------------------------------------------------------
# per-launch profiling hook (bench.py): HIP events on the stream the kernels are launched on
# ------------------------------------------------------------------------------------------------
class LaunchProfiler:
    """When installed (``ops.PROFILER = LaunchProfiler()``), every dz_conv2d_forward / dz_spconv_forward
    call is bracketed by a pair of timing events recorded on the launch stream; ``summary()`` returns
    per-kernel-variant launch counts, total device time and algorithmic FLOPs / bytes."""

    def __init__(self):
        self.records = []

    def wrap(self, name, flops, nbytes, fn):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        st = torch.cuda.current_stream()
        e0.record(st)
        fn()
        e1.record(st)
        self.records.append((name, flops, nbytes, e0, e1))

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, flops, nbytes, e0, e1 in self.records:
            a = agg.setdefault(name, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0})
            a['launches'] += 1
            a['ms'] += e0.elapsed_time(e1)
            a['flops'] += flops
            a['bytes'] += nbytes
        return agg


PROFILER = None


# ------------------------------------------------------------------------------------------------
# device guard: every wrapper above allocates its outputs / workspaces with the current device and launches on its current
# stream, so each public entry runs under torch.cuda.device(<device of its first tensor argument>) - tensors living on a GPU
# other than the current one (one process driving several GPUs) then get their kernels on their own device and stream
# ------------------------------------------------------------------------------------------------
def _guarded(fn):
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        dev = None
        for a in list(args) + list(kwargs.values()):
            if torch.is_tensor(a) and a.is_cuda:
                dev = a.device
                break
            if isinstance(a, SparseLevel):
                dev = a.coords.device
                break
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapper


def _install_guards():
    import types
    g = globals()
    for name, obj in list(g.items()):
        if isinstance(obj, types.FunctionType) and obj.__module__ == __name__ and not name.startswith('_') and name not in ('grid_size_of', 'math_id', 'pair16_pack', 'pair16_unpack'):
            g[name] = _guarded(obj)
    for name in ('build_from_coords', 'downsample', 'neighbors_to'):
        setattr(SparseLevel, name, _guarded(getattr(SparseLevel, name)))


_install_guards()
