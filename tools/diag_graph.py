"""Diagnostic: which stage of the frame pipeline misbehaves under HIP-graph replay (bisect by stage)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detzero_amd import ops
from detzero_amd.centerpoint import FramePipeline, synth_detector
from detzero_amd.synth import VOXEL_SIZE_02
from tests.util import masked_frame


def P(*a):
    print(*a, flush=True)


dev = torch.device('cuda', 0)
model, cfg, info = synth_detector(VOXEL_SIZE_02, seed=2)
model = model.to(dev)
pipe = FramePipeline(model, info)
frames = [torch.from_numpy(masked_frame(s, 20000)).to(dev) for s in (3, 4, 5)]
nmax = max(f.shape[0] for f in frames)
static_in = torch.zeros((nmax, 5), device=dev)


def load(f):
    static_in.zero_(); static_in[:, 0] = 1e6; static_in[:f.shape[0]] = f


def stage_fn(upto):
    rng = info.point_cloud_range
    def fn(points):
        voxels, zyx, nump, d_n = ops.voxelize_hard_nosync(points, rng, info.voxel_size, 5, 200000, xy_range_mask=True)
        feats = ops.mean_vfe(voxels, nump, d_m=d_n)
        if upto == 'vox':
            return feats, d_n
        coords = torch.cat([zyx.new_zeros((zyx.shape[0], 1)), zyx], dim=1).contiguous()
        res = model.backbone3d.run(feats, coords, 1, d_n)
        x, lvl = res['encoded']
        if upto == 'bb3d':
            return x, lvl.d_m
        bev = ops.sparse_to_bev(x, lvl, x.shape[1], pad=1)
        concat = model.backbone2d.run(bev, 1)
        if upto == 'bev':
            return concat, lvl.d_m
        head, h, w = model.dense_head.run_convs(concat, 1)
        if upto == 'head':
            return head, lvl.d_m
        boxes, scores, labels, counts = ops.centerhead_decode(
            head, h, w, 3, 500, 0.03, [-80, -80, -10.0, 80, 80, 10.0], info.point_cloud_range, info.voxel_size, 8)
        if upto == 'decode':
            return boxes, counts
        keep, d_nk = ops.nms_rotated_nosync(boxes[0], counts[0:1], 0.7, 500)
        return keep, d_nk
    return fn


only = sys.argv[1:] or ['vox', 'bb3d', 'bev', 'head', 'decode', 'nms', 'full']
for name in only:
    fn = pipe if name == 'full' else stage_fn(name)
    eager = []
    for f in frames:
        load(f)
        o, n = fn(static_in); torch.cuda.synchronize()
        eager.append((o.clone(), n.clone()))
    load(frames[0])
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(static_in)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        g_o, g_n = fn(static_in)
    torch.cuda.synchronize()
    ok = True
    for rep in range(2):
        for i, f in enumerate(frames):
            load(f); torch.cuda.synchronize()
            g.replay(); torch.cuda.synchronize()
            k = int(eager[i][1].reshape(-1)[0].item())
            same_n = torch.equal(g_n, eager[i][1])
            same_o = torch.equal(g_o[:k], eager[i][0][:k]) if g_o.dim() > 0 and name not in ('bev', 'head') else torch.equal(g_o, eager[i][0])
            ok = ok and same_n and same_o
            P('stage', name, 'rep', rep, 'frame', i, 'count', k, 'same_count', same_n, 'same_out', same_o)
    P('STAGE', name, 'OK' if ok else 'MISMATCH')
    del g
P('graph diag done')
