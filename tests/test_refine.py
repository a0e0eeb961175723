"""Refining module (GRM / PRM): state-dict compatibility with the reference (CPU) and parity of the HIP path
with the outputs of the reference's own GeometryTransformer / PositionTransformer (GPU)."""
import os

import numpy as np
import pytest
import torch

from detzero_amd.config import AttrDict
from detzero_amd.synth import synth_state_dict

GCFG = AttrDict({'QUERY_ENCODER': [128, 128], 'MEMORY_ENCODER': [128, 128], 'REGRESSION_MLP': [512], 'EMBED_DIMS': 256,
                 'ANCHOR_SIZES': [[4.8, 1.8, 1.5], [10.0, 2.6, 3.2], [2.0, 1.0, 1.6]],
                 'DECODER': {'NAME': 'GeometryHead', 'num_classes': 3, 'num_heads': 8, 'num_decoder_layers': 1,
                             'auxiliary': True, 'cross_only': False, 'memory_self_attn': False, 'hidden_channel': 256,
                             'ffn_channel': 256, 'dropout': 0.1, 'bn_momentum': 0.1, 'activation': 'relu'}})
PCFG = AttrDict({'QUERY_ENCODER': [128, 128], 'MEMORY_ENCODER': [128, 128], 'REGRESSION_MLP': [512],
                 'LOSS_CLS': {'type': 'CrossEntropyLoss'},
                 'DECODER': {'NAME': 'PositionHead', 'num_classes': 3, 'num_heads': 8, 'num_decoder_layers': 1,
                             'auxiliary': True, 'cross_only': False, 'hidden_channel': 256, 'dropout': 0.1,
                             'bn_momentum': 0.1, 'activation': 'relu', 'ffn_channel': 256}})


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'refine_golden.npz'))


def _models():
    from detzero_amd.refine_modules import GeometryTransformer, PositionTransformer
    return GeometryTransformer(GCFG, query_point_dims=11, memory_point_dims=4).eval(), \
        PositionTransformer(PCFG, query_point_dims=32, memory_point_dims=32).eval()


def test_state_dict_identical_to_reference_manifest(g):
    """Key names AND shapes equal the reference modules' own state_dict (recorded when the golden was made)."""
    grm, prm = _models()
    for model, tag in ((grm, 'grm'), (prm, 'prm')):
        ref = dict(zip(g[tag + '_keys'].tolist(), g[tag + '_shapes'].tolist()))
        mine = {k: str(tuple(v.shape)) for k, v in model.state_dict().items()}
        assert mine == ref, (sorted(set(mine) ^ set(ref))[:6], [(k, mine[k], ref[k]) for k in mine if k in ref and mine[k] != ref[k]][:4])
        model.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1), strict=True)


@pytest.mark.gpu
@pytest.mark.parametrize('math', ['f32', 'f16x2', 'bf16x2'])
def test_grm_matches_reference(device, g, math, monkeypatch):
    from detzero_amd import refine_modules
    monkeypatch.setattr(refine_modules, 'SPLIT_MIN_ROWS', 1)          # the fixture is small: force the split stacks on
    grm, _ = _models()
    grm.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in grm.state_dict().items()}, seed=5), strict=True)
    grm = grm.to(device).set_math(math)
    data = {k[len('grm_in_'):]: torch.from_numpy(g[k]).to(device) for k in g.files if k.startswith('grm_in_')}
    res = grm(data)
    torch.testing.assert_close(res['memory'].cpu(), torch.from_numpy(g['grm_memory']), rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(res['query'].cpu(), torch.from_numpy(g['grm_query']), rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(grm.preds_dict['geometry_cls'].cpu(), torch.from_numpy(g['grm_cls']), rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(grm.preds_dict['geometry_reg'].cpu(), torch.from_numpy(g['grm_reg']), rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(res['batch_box_preds'].cpu(), torch.from_numpy(g['grm_boxes']), rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('math', ['f32', 'f16x2', 'bf16x2'])
def test_prm_matches_reference(device, g, math, monkeypatch):
    from detzero_amd import refine_modules
    monkeypatch.setattr(refine_modules, 'SPLIT_MIN_ROWS', 1)
    _, prm = _models()
    prm.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in prm.state_dict().items()}, seed=6), strict=True)
    prm = prm.to(device).set_math(math)
    data = {k[len('prm_in_'):]: torch.from_numpy(g[k]).to(device) for k in g.files if k.startswith('prm_in_')}
    res = prm(data)
    torch.testing.assert_close(res['query'].cpu(), torch.from_numpy(g['prm_query']), rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(res['memory'].cpu()[:, :, ::16], torch.from_numpy(g['prm_memory']), rtol=1e-3, atol=1e-4)
    valid = torch.from_numpy(g['prm_in_padding_mask']) == 0          # padded queries attend to nothing meaningful
    for k in ('center_reg', 'heading_cls', 'heading_reg'):
        torch.testing.assert_close(prm.preds_dict[k].cpu()[valid], torch.from_numpy(g['prm_' + k])[valid], rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(res['batch_box_preds'].cpu()[valid], torch.from_numpy(g['prm_boxes'])[valid], rtol=1e-3, atol=1e-3)


@pytest.mark.gpu
def test_group_max_and_layernorm(device):
    from detzero_amd import ops
    gen = torch.Generator().manual_seed(0)
    x = torch.randn((5 * 37, 192), generator=gen)
    out = ops.group_max(x.to(device), 5, 37).cpu()
    assert torch.equal(out, x.view(5, 37, 192).max(dim=1)[0])
    a = torch.randn((77, 256), generator=gen); b = torch.randn((77, 256), generator=gen)
    gm = torch.rand(256, generator=gen) + 0.5; bt = torch.randn(256, generator=gen)
    ln = ops.add_layernorm(a.to(device), b.to(device), gm.to(device), bt.to(device), 1e-5).cpu()
    ref = torch.nn.functional.layer_norm(a + b, (256,), gm, bt, 1e-5)
    torch.testing.assert_close(ln, ref, rtol=1e-5, atol=1e-5)
    s = ops.add_layernorm(a.to(device), b.to(device), None, None, norm=False).cpu()
    assert torch.equal(s, a + b)


@pytest.mark.gpu
def test_linear_over_the_2gib_buffer_window(device):
    """Inputs larger than the 32-bit buffer-offset window are processed in row chunks (group addend aligned)."""
    from detzero_amd import ops
    rows, cin, cout, grp = 9_000_000, 64, 16, 1000                  # 2.3 GB input
    x = torch.empty((rows, cin), dtype=torch.float32, device=device)
    x.copy_(torch.arange(rows, device=device, dtype=torch.float32)[:, None] % 7.0 - 3.0)
    x[:, 1] = 1.0
    w = torch.zeros((cin, cout), device=device); w[0, 0] = 1.0; w[1, 1] = 2.0
    gs = torch.arange(rows // grp, device=device, dtype=torch.float32)[:, None].repeat(1, cout).contiguous()
    one = torch.ones(cout, device=device); zero = torch.zeros(cout, device=device)
    y = ops.linear(x, w, one, zero, False, cout, group_shift=gs, group_rows=grp)
    idx = torch.tensor([0, 999, 1000, 8_388_607, 8_388_608, rows - 1], device=device)
    exp0 = (idx.float() % 7.0 - 3.0) + (idx // grp).float()
    exp1 = 2.0 + (idx // grp).float()
    assert torch.equal(y[idx, 0], exp0) and torch.equal(y[idx, 1], exp1)


@pytest.mark.gpu
@pytest.mark.parametrize('n,t,bev', [(60000, 40, False), (180000, 128, True), (1000, 1, False), (5000, 70, False)])
def test_object_crop_with_compaction_vs_oracle(device, n, t, bev):
    """Refine data path, crop step (prepare_object_data.py:250-273,310): per-object point arrays equal the oracle's
    `pts[mask[i]]` bit for bit (float64 rows, order included) - without the dense (T,M) mask."""
    from detzero_amd import object_crop
    from detzero_amd.synth import synth_boxes, synth_waymo_frame
    from oracle import crop as ocrop
    rng = np.random.default_rng(n + t)
    f = synth_waymo_frame(n % 97, n)
    pts = np.concatenate([f[:, :5], np.where(rng.random(n) < 0.1, 1.0, -1.0)[:, None]], axis=1).astype(np.float32)    # NLZ column
    a = rng.uniform(-np.pi, np.pi)
    pose = np.eye(4)
    pose[:2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
    pose[:3, 3] = rng.uniform(-50, 50, size=3)
    boxes_l = synth_boxes(t, t, 60.0).astype(np.float64)
    boxes_l[:, 2] = rng.uniform(-0.5, 1.0, size=t)
    from detzero_amd.track_adapter import transform_boxes3d
    boxes_g = transform_boxes3d(boxes_l.copy(), pose)
    ref = ocrop.crop_frame_objects(pts, pose, boxes_g, 1.1, bev)
    got = object_crop.crop_frame_objects(pts, pose, boxes_g, 1.1, bev, device=device)
    assert len(got) == len(ref) == t
    assert t < 10 or sum(r.shape[0] for r in ref) > 0
    for g, r in zip(got, ref):
        assert g.dtype == np.float64 and g.shape == r.shape and np.array_equal(g, r)


@pytest.mark.gpu
def test_object_crop_edge_cases(device):
    from detzero_amd import object_crop, ops
    assert object_crop.crop_objects(np.zeros((10, 4)), np.zeros((0, 7)), device) == []
    out = object_crop.crop_objects(np.zeros((0, 4)), np.ones((3, 7)), device)
    assert len(out) == 3 and all(o.shape == (0, 4) for o in out)
    # overlapping boxes keep a point once per box; capacity overflow is reported through d_total
    pts = np.zeros((100, 4)); pts[:, 3] = np.arange(100)
    boxes = np.tile(np.array([[0, 0, 0, 2, 2, 2, 0.3]]), (3, 1))
    out = object_crop.crop_objects(pts, boxes, device)
    assert all(np.array_equal(o[:, 3], np.arange(100)) for o in out)
    with pytest.raises(ValueError):
        object_crop.crop_objects(pts, boxes, device, cap=250)
