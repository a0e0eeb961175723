"""How much of a wave-private tile's gathers is shared along x (CPU, oracle tables of one synthetic 160k-point frame).

Rows are kept in key order, so the three x taps of a window row are consecutive input rows and the neighbours a tile of 32 consecutive
output rows needs at one (tz, ty) offset lie in ONE short range of input rows.  For every level: the rows gathered today (one per
(tap, row) pair), the distinct rows per (tile, tz, ty) group, and the rows a contiguous load of each group's range would fetch
(groups whose range exceeds the tile's rows + 64 counted as gathered per tap).  DESIGN.md section 8, "next".
usage: python tools/xrun_stats.py"""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
from oracle import sparse as osp, voxelize as ov
from detzero_amd.synth import synth_waymo_frame, POINT_CLOUD_RANGE, VOXEL_SIZE_01
pts = synth_waymo_frame(5, 160000)
_, c, _ = ov.hard_voxelize(pts[ov.mask_points_by_range(pts, POINT_CLOUD_RANGE)], POINT_CLOUD_RANGE, VOXEL_SIZE_01, 5, 200000)
coords = np.concatenate([np.zeros((c.shape[0],1),np.int32), c],1)
grid = ov.grid_size_of(POINT_CLOUD_RANGE, VOXEL_SIZE_01)
shape = [int(grid[2])+1, int(grid[1]), int(grid[0])]
coords = coords[osp.canonical_order(coords, shape)]
K3=(3,3,3)
def stats(name, cin_coords, in_shape, out_coords, k, s, p, tile=32):
    tab = osp.neighbor_table(cin_coords, in_shape, out_coords, k, s, p).astype(np.int64)
    m = out_coords.shape[0]
    pairs = (tab>=0).sum()
    t3 = tab.reshape(9,3,m)
    tot_range = 0; tot_unique = 0; big = 0; groups = 0; hist=[]
    for t0 in range(0, m, tile):
        blk = t3[:, :, t0:t0+tile]
        for g in range(9):
            v = blk[g][blk[g]>=0]
            if v.size == 0: continue
            groups += 1
            rng = v.max()-v.min()+1
            u = np.unique(v).size
            tot_unique += u
            hist.append(rng)
            if rng > tile + 64: big += 1; tot_range += v.size   # fallback: per-tap gathers
            else: tot_range += rng
    hist=np.array(hist)
    print('%s: rows %d pairs/row %.2f | gathers now %d rows | unique per group sum %d (%.2fx fewer) | range-load rows %d (%.2fx fewer), groups over tile+64 rows: %.1f%%, median range %d, p90 %d' % (
        name, m, pairs/m, pairs, tot_unique, pairs/tot_unique, tot_range, pairs/tot_range, 100.0*big/max(groups,1), np.median(hist), np.percentile(hist,90)))
stats('L1 subm (32-row tiles)', coords, shape, coords, K3, (1,1,1), (1,1,1))
oc, osh = osp.conv_out_coords(coords, shape, K3, (2,2,2), (1,1,1))
stats('L1->L2 down (32)', coords, shape, oc, K3, (2,2,2), (1,1,1))
stats('L2 subm (32)', oc, list(osh), oc, K3, (1,1,1), (1,1,1))
oc3, osh3 = osp.conv_out_coords(oc, list(osh), K3, (2,2,2), (1,1,1))
stats('L2->L3 down (256)', oc, list(osh), oc3, K3, (2,2,2), (1,1,1), tile=256)
for t in (128, 256):
    stats('L3 subm (%d)' % t, oc3, list(osh3), oc3, K3, (1,1,1), (1,1,1), tile=t)
oc4, osh4 = osp.conv_out_coords(oc3, list(osh3), K3, (2,2,2), (0,1,1))
stats('L3->L4 down (128)', oc3, list(osh3), oc4, K3, (2,2,2), (0,1,1), tile=128)
for t in (128, 256):
    stats('L4 subm (%d)' % t, oc4, list(osh4), oc4, K3, (1,1,1), (1,1,1), tile=t)
