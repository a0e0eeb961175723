#!/usr/bin/env python
"""Pin the oracle on the REAL spconv - for a maintainer whose environment has it (this build container does not, and there
is no network; see DESIGN.md section 3 "parity unpinned").

    python tools/gen_spconv_golden.py            # needs spconv 2.x (+cumm) and a CUDA/ROCm device spconv supports
      -> tests/golden/spconv_golden.npz          # picked up by tests/test_oracle_golden.py::test_spconv_golden_if_present

What it records, on seeded synthetic inputs (detzero_amd.synth, no data files):
  * the hard voxelizer exactly as the reference calls it (datasets/processor/data_processor.py:61-91):
    spconv.utils.Point2VoxelCPU3d(...).point_to_voxel(tv.from_numpy(points)) -> voxels, coordinates (zyx), num_points,
    for a 20k-point frame (0.2 m voxels) and for a max_voxels that binds;
  * SubMConv3d / SparseConv3d forward (the geometries of backbone3d.py:243-280: 3x3x3 subm, 3x3x3 stride 2 pad 1,
    3x3x3 stride 2 pad (0,1,1), (3,1,1) stride (2,1,1) pad 0) on a small sparse tensor, with the module's own weight tensor
    (whatever layout the installed spconv uses - the test detects it from the shape) and the output indices / features.
spconv's output ROW ORDER is an implementation detail (hash-table order); the test compares after sorting rows by the linear
voxel key, which is what "bit-exact indices / rulebook" means throughout this repository.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import spconv.pytorch as spconv
    from cumm import tensorview as tv
    from spconv.utils import Point2VoxelCPU3d
    from detzero_amd.synth import POINT_CLOUD_RANGE, VOXEL_SIZE_02, synth_waymo_frame
    out = {}
    pts = synth_waymo_frame(5, 20000)
    m = (pts[:, 0] >= POINT_CLOUD_RANGE[0]) & (pts[:, 0] <= POINT_CLOUD_RANGE[3]) & (pts[:, 1] >= POINT_CLOUD_RANGE[1]) & (pts[:, 1] <= POINT_CLOUD_RANGE[4])
    pts = np.ascontiguousarray(pts[m])
    out['vox_points'] = pts
    for tag, max_voxels in (('full', 200000), ('bind', 3000)):
        gen = Point2VoxelCPU3d(vsize_xyz=VOXEL_SIZE_02, coors_range_xyz=POINT_CLOUD_RANGE.tolist(), num_point_features=5,
                               max_num_points_per_voxel=5, max_num_voxels=max_voxels)
        v, c, n = gen.point_to_voxel(tv.from_numpy(pts))
        out['vox_%s_voxels' % tag], out['vox_%s_coords' % tag], out['vox_%s_num' % tag] = v.numpy().copy(), c.numpy().copy(), n.numpy().copy()
        out['vox_%s_max' % tag] = np.array(max_voxels)
    rng = np.random.default_rng(11)
    shape, batch, n_act, cin, cout = [9, 24, 26], 2, 1500, 16, 32
    cells = batch * shape[0] * shape[1] * shape[2]
    lin = rng.choice(cells, size=n_act, replace=False)
    per = shape[0] * shape[1] * shape[2]
    coords = np.stack([lin // per, (lin % per) // (shape[1] * shape[2]), (lin // shape[2]) % shape[1], lin % shape[2]], 1).astype(np.int32)
    feats = rng.standard_normal((n_act, cin)).astype(np.float32)
    out['conv_coords'], out['conv_feats'], out['conv_shape'], out['conv_batch'] = coords, feats, np.array(shape), np.array(batch)
    dev = torch.device('cuda')
    geoms = {'subm': ('subm', 3, 1, 1), 'down': ('conv', 3, 2, 1), 'down011': ('conv', 3, 2, (0, 1, 1)), 'out311': ('conv', (3, 1, 1), (2, 1, 1), 0)}
    for tag, (kind, k, s, p) in geoms.items():
        torch.manual_seed(3)
        mod = (spconv.SubMConv3d(cin, cout, k, padding=p, bias=False, indice_key=tag) if kind == 'subm'
               else spconv.SparseConv3d(cin, cout, k, stride=s, padding=p, bias=False, indice_key=tag)).to(dev)
        x = spconv.SparseConvTensor(torch.from_numpy(feats).to(dev), torch.from_numpy(coords).to(dev), shape, batch)
        with torch.no_grad():
            y = mod(x)
        out['conv_%s_weight' % tag] = mod.weight.detach().cpu().numpy()
        out['conv_%s_indices' % tag] = y.indices.cpu().numpy().astype(np.int32)
        out['conv_%s_features' % tag] = y.features.detach().cpu().numpy()
        out['conv_%s_shape' % tag] = np.array(y.spatial_shape)
    path = os.path.join(ROOT, 'tests', 'golden', 'spconv_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, 'spconv', getattr(spconv, '__version__', '?'))


if __name__ == '__main__':
    main()
