"""CPU: the drop-in boundary - the C-ABI library exports every symbol include/detzero_hip.h
declares, the product package never touches the oracle, state-dict layout, config plumbing."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'detzero_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(dz_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from detzero_amd import lib as L
    from detzero_amd.build import build
    build(verbose=False)
    syms = _header_symbols()
    assert len(syms) >= 25
    cdll = ctypes.CDLL(L.LIB_PATH)
    for s in syms:
        assert hasattr(cdll, s), 'missing export %s' % s
    assert sorted(L.exported_symbols()) == syms            # binding table == header
    L.load()
    assert b'gfx950' in L.load().dz_version()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from detzero_amd import lib as L
    monkeypatch.setattr(L, '_lib', None)
    monkeypatch.setattr(L, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(L.DetZeroHipError):
        L.load()


def test_cpu_tensors_are_refused():
    from detzero_amd import ops
    from detzero_amd.lib import DetZeroHipError
    with pytest.raises(DetZeroHipError):
        ops.mean_vfe(torch.zeros(4, 5, 5), torch.ones(4, dtype=torch.int32))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'detzero_amd')
    bad = []
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dp, f), errors='ignore').read()
                if re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M) or 'oracle/' in txt and f.endswith('.py'):
                    bad.append(f)
    assert not bad, bad


def _model(voxel=(0.1, 0.1, 0.15)):
    from detzero_amd.centerpoint import SyntheticDatasetInfo, build_network
    from detzero_amd.config import centerpoint_1sweep_cfg
    cfg = centerpoint_1sweep_cfg(voxel)
    info = SyntheticDatasetInfo(cfg)
    torch.manual_seed(0)
    return build_network(cfg.MODEL, len(cfg.CLASS_NAMES), info), cfg, info


def test_state_dict_layout_matches_reference():
    """Key names and shapes of SURVEY.md §5 (reference checkpoints load with load_params_from_file)."""
    model, cfg, info = _model()
    sd = model.state_dict()
    assert tuple(info.grid_size) == (1504, 1504, 40)
    assert model.backbone3d.sparse_shape == [41, 1504, 1504]
    expect = {
        'global_step': (1,),
        'backbone3d.conv_input.0.weight': (16, 3, 3, 3, 5),
        'backbone3d.conv_input.1.running_var': (16,),
        'backbone3d.conv1.0.conv1.weight': (16, 3, 3, 3, 16), 'backbone3d.conv1.0.conv1.bias': (16,),
        'backbone3d.conv1.1.bn2.num_batches_tracked': (),
        'backbone3d.conv2.0.0.weight': (32, 3, 3, 3, 16), 'backbone3d.conv2.0.1.weight': (32,),
        'backbone3d.conv3.2.conv2.weight': (64, 3, 3, 3, 64),
        'backbone3d.conv4.0.0.weight': (128, 3, 3, 3, 64),
        'backbone3d.conv_out.0.weight': (128, 3, 1, 1, 128), 'backbone3d.conv_out.1.bias': (128,),
        'backbone2d.blocks.0.1.weight': (128, 256, 3, 3), 'backbone2d.blocks.0.2.running_mean': (128,),
        'backbone2d.blocks.0.16.weight': (128, 128, 3, 3), 'backbone2d.blocks.1.1.weight': (256, 128, 3, 3),
        'backbone2d.deblocks.0.0.weight': (128, 256, 1, 1), 'backbone2d.deblocks.1.0.weight': (256, 256, 2, 2),
        'dense_head.shared_conv.0.weight': (64, 512, 3, 3), 'dense_head.shared_conv.0.bias': (64,),
        'dense_head.heads_list.0.center.0.0.weight': (64, 64, 3, 3), 'dense_head.heads_list.0.center.0.0.bias': (64,),
        'dense_head.heads_list.0.center.0.1.running_var': (64,), 'dense_head.heads_list.0.center.1.weight': (2, 64, 3, 3),
        'dense_head.heads_list.0.hm.1.bias': (3,), 'dense_head.heads_list.0.iou.1.weight': (1, 64, 3, 3),
        'dense_head.heads_list.0.dim.1.weight': (3, 64, 3, 3), 'dense_head.heads_list.0.rot.1.bias': (2,),
        'dense_head.heads_list.0.center_z.1.bias': (1,),
    }
    for k, shp in expect.items():
        assert k in sd, k
        assert tuple(sd[k].shape) == shp, (k, tuple(sd[k].shape))
    assert not any('conv_input.0.bias' in k or 'conv2.0.0.bias' in k for k in sd)
    n3d = sum(v.numel() for k, v in sd.items() if k.startswith('backbone3d') and 'weight' in k and v.dim() == 5)
    assert 2_600_000 < n3d < 2_800_000          # backbone3d ~2.7 M params
    assert float(sd['dense_head.heads_list.0.hm.1.bias'][0]) == pytest.approx(-2.19)


def test_state_dict_identical_to_reference_manifest():
    """EVERY key and shape equals what the reference's own classes register for centerpoint_1sweep (tests/golden/det_manifest.json,
    recorded by gen_det_manifest.py from backbone3d.py / backbone2d.py / center_head.py), and a strict load of it succeeds."""
    import json
    from detzero_amd.synth import synth_state_dict
    model, cfg, info = _model()
    ref = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'det_manifest.json')))
    mine = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert mine == ref, (sorted(set(mine) ^ set(ref))[:6], [(k, mine[k], ref[k]) for k in mine if k in ref and mine[k] != ref[k]][:4])
    assert len(ref) == 288
    model.load_state_dict(synth_state_dict({k: tuple(v) for k, v in ref.items()}, seed=2), strict=True)


def test_registry_and_processor_names():
    from detzero_amd import det_modules
    from detzero_amd.data_processor import DataProcessor
    for name in ('MeanVFE', 'DynamicMeanVFE', 'VoxelResBackBone8x', 'HeightCompression', 'BaseBEVBackbone', 'CenterHead'):
        assert name in det_modules.__all__
    _, cfg, info = _model()
    dp = DataProcessor(cfg.DATA_CONFIG.DATA_PROCESSOR, info.point_cloud_range, training=False, num_point_features=5)
    assert tuple(dp.grid_size) == (1504, 1504, 40) and len(dp.data_processor_queue) == 3
    pts = np.array([[0, 0, 0, 0, 0], [80, 0, 0, 0, 0], [75.2, -75.2, 9, 0, 0]], np.float32)
    out = dp.mask_points_and_boxes_outside_range({'points': pts}, config=cfg.DATA_CONFIG.DATA_PROCESSOR[0])
    assert out['points'].shape[0] == 2          # xy-only, inclusive bounds


def test_training_mode_refused():
    from detzero_amd.lib import DetZeroHipError
    model, _, _ = _model()
    model.train()
    with pytest.raises(DetZeroHipError):
        model({'batch_size': 1})


def test_config_yaml_include(tmp_path):
    from detzero_amd.config import cfg_from_yaml_file
    base = tmp_path / 'cfgs' / 'det_dataset_cfgs'
    base.mkdir(parents=True)
    (base / 'waymo.yaml').write_text('POINT_CLOUD_RANGE: [-75.2, -75.2, -2, 75.2, 75.2, 4.0]\nDATASET: X\n')
    mdir = tmp_path / 'cfgs' / 'det_model_cfgs'
    mdir.mkdir()
    (mdir / 'm.yaml').write_text('CLASS_NAMES: [A]\nDATA_CONFIG:\n    _BASE_CONFIG_: cfgs/det_dataset_cfgs/waymo.yaml\n'
                                 '    DATASET: Y\nMODEL:\n    NAME: CenterPoint\n')
    cfg = cfg_from_yaml_file(str(mdir / 'm.yaml'), base_dir=str(tmp_path))
    assert cfg.DATA_CONFIG.DATASET == 'Y' and cfg.DATA_CONFIG.POINT_CLOUD_RANGE[0] == -75.2 and cfg.MODEL.NAME == 'CenterPoint'


def test_collate_batch_matches_reference_layout():
    from detzero_amd.dataset_utils import collate_batch
    a = {'voxels': np.zeros((3, 5, 5), np.float32), 'voxel_coords': np.ones((3, 3), np.int32), 'voxel_num_points': np.ones(3, np.int32),
         'points': np.zeros((7, 5), np.float32), 'gt_boxes': np.ones((2, 8), np.float32), 'frame_id': 4}
    b = {'voxels': np.zeros((2, 5, 5), np.float32), 'voxel_coords': np.ones((2, 3), np.int32), 'voxel_num_points': np.ones(2, np.int32),
         'points': np.zeros((4, 5), np.float32), 'gt_boxes': np.ones((5, 8), np.float32), 'frame_id': 5}
    out = collate_batch([a, b])
    assert out['batch_size'] == 2 and out['voxels'].shape == (5, 5, 5) and out['voxel_coords'].shape == (5, 4)
    assert out['voxel_coords'][:, 0].tolist() == [0, 0, 0, 1, 1] and out['points'].shape == (11, 6)
    assert out['gt_boxes'].shape == (2, 5, 8) and out['gt_boxes'][0, 2:].sum() == 0 and out['frame_id'].tolist() == [4, 5]


def test_math_modes_and_their_tensor_encoding():
    """'f16' computes on the tensors of 'f16x2' (same packers, same conversion entry points); the header names the same ids."""
    from detzero_amd import ops
    assert ops.MATH_MODES == {'f32': 0, 'f16x2': 1, 'bf16x2': 2, 'f16': 3}
    assert [ops.storage_math(m) for m in (0, 1, 2, 3)] == [0, 1, 2, 1]
    hdr = open(os.path.join(ROOT, 'include', 'detzero_hip.h')).read()
    for name, val in (('DZ_MATH_F32', 0), ('DZ_MATH_F16X2', 1), ('DZ_MATH_BF16X2', 2), ('DZ_MATH_F16', 3)):
        assert ('#define %s %d' % (name, val)) in hdr
    x = torch.randn(4, 16)
    assert torch.equal(ops.pair16_pack(x, 3), ops.pair16_pack(x, 1))
    with pytest.raises(Exception):
        ops.math_id('fp8')


def test_capability_queries_answer_without_a_gpu():
    """The `*_supported` / workspace entry points of the fused refiner and PDV kernels are host code: they must answer on a machine
    without a GPU (the Python side uses them to choose between the fused kernel and the layer-by-layer path)."""
    from detzero_amd import lib as L
    lib = L.load()
    # folded cross-attention: E = 256, heads * queries <= 32
    assert lib.dz_xattn_folded_supported(3, 256, 8) == 1
    assert lib.dz_xattn_folded_supported(4, 256, 8) == 1
    assert lib.dz_xattn_folded_supported(5, 256, 8) == 0          # 40 folded rows
    assert lib.dz_xattn_folded_supported(3, 128, 4) == 0
    assert lib.dz_xattn_folded_supported(200, 256, 8) == 0        # PRM: keeps K / V + the attention core
    assert lib.dz_xattn_folded_workspace_bytes(128, 4096) >= 128 * 32 * 256 * 4
    # PDV grid pooling: the two instances of the centerpoint_pdv configs
    assert lib.dz_pdv_sa_pool_supported(64, 80, 32, 32, 16, 1, 1) == 1
    assert lib.dz_pdv_sa_pool_supported(128, 144, 64, 64, 16, 1, 1) == 1
    assert lib.dz_pdv_sa_pool_supported(128, 144, 64, 64, 32, 1, 1) == 0
    assert lib.dz_pdv_sa_pool_supported(128, 144, 64, 64, 16, 1, 0) == 0
    assert lib.dz_pdv_sa_pool_supported(126, 144, 64, 64, 16, 1, 1) == 0
    # PDV part counts on a BEV grid of RoI lists: per frame a 64-byte header, 64 x 64 lines of 16 ints, the staged box table (10 floats
    # per RoI, padded to whole lines)
    assert lib.dz_pdv_part_counts_ws_bytes(8, 487) == 8 * (16 + 64 * 64 * 16 + 4880) * 4
    assert lib.dz_pdv_part_counts_ws_bytes(1, 0) == (16 + 64 * 64 * 16) * 4
    assert lib.dz_pdv_part_counts_ws_bytes(0, 5) == 0
    assert lib.dz_pdv_part_counts_ws_bytes(2, 3) % 64 == 0
