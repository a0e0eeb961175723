"""Test-time augmentation: the oracle against the reference's own code (CPU), the device path against both (GPU)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import wbf as oracle_wbf

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import gen_tta_golden as gen          # noqa: E402  (config and case list of the fixture; importing it does not touch the reference)

TAGS = ['f%d_' % seed for seed, _ in gen.FRAMES]


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'tta_golden.npz'))


# ------------------------------------------------------------------------------------------------ CPU
def test_augmentor_names_and_oracle_points(g):
    from detzero_amd.tta import TestTimeAugmentor, parse_op
    aug = TestTimeAugmentor(gen.AUG_CONFIG)
    ops = g['ops'].tolist()
    assert aug.op_names == ops and len(ops) == 15
    assert parse_op('tta_rot_-0.78539816') == (4, -0.78539816) and parse_op('tta_flip_xy') == (3, 0.0)
    for op in ops:          # float32 matmul in torch vs numpy: equal up to 1 ulp at 75 m
        np.testing.assert_allclose(oracle_wbf.augment_points(g['points'], op), g['points_' + op], rtol=0, atol=1e-5)


@pytest.mark.parametrize('tag', TAGS)
def test_oracle_restore_and_fusion_equal_reference(g, tag):
    ops = g['ops'].tolist()
    np.testing.assert_allclose(oracle_wbf.restore_boxes(g[tag + 'pred_boxes'], ops), g[tag + 'restored'], rtol=0, atol=1e-5)
    b, s, l = oracle_wbf.weighted_boxes_fusion_3d(g[tag + 'restored'], g[tag + 'pred_scores'], g[tag + 'pred_labels'])
    np.testing.assert_array_equal(b, g[tag + 'fused_boxes'])          # bit for bit, float64
    np.testing.assert_array_equal(s, g[tag + 'fused_scores'])
    np.testing.assert_array_equal(l, g[tag + 'fused_labels'])


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_augment_points_match_reference(device, g):
    from detzero_amd.tta import TestTimeAugmentor
    aug = TestTimeAugmentor(gen.AUG_CONFIG)
    out = aug.augment(torch.from_numpy(g['points']).to(device)).cpu().numpy()
    for i, op in enumerate(aug.op_names):
        np.testing.assert_allclose(out[i], g['points_' + op], rtol=0, atol=1e-5, err_msg=op)
    d = aug.forward({'points': torch.from_numpy(g['points']).to(device), 'frame_id': 3})
    assert list(d) == aug.op_names and d['tta_flip_x']['frame_id'] == 3


@pytest.mark.gpu
@pytest.mark.parametrize('tag', TAGS)
def test_restore_and_fusion_match_reference(device, g, tag):
    from detzero_amd import tta
    ops = g['ops'].tolist()
    pb = torch.from_numpy(g[tag + 'pred_boxes']).to(device)
    restored = tta.restore_boxes(pb.clone()[None].contiguous(), ops)[0]
    np.testing.assert_allclose(restored.cpu().numpy(), g[tag + 'restored'], rtol=0, atol=1e-5)
    # fusion of the reference's restored boxes: same clusters, hence the same numbers
    b, s, l = tta.wbf_online(torch.from_numpy(g[tag + 'restored']).to(device), torch.from_numpy(g[tag + 'pred_scores']).to(device)[..., None],
                             torch.from_numpy(g[tag + 'pred_labels']).to(device)[..., None])
    assert b.dtype == torch.float64 and s.dtype == torch.float64 and l.dtype == torch.int64
    assert b.shape[0] == g[tag + 'fused_boxes'].shape[0]
    np.testing.assert_array_equal(l.cpu().numpy(), g[tag + 'fused_labels'])
    np.testing.assert_allclose(s.cpu().numpy(), g[tag + 'fused_scores'], rtol=0, atol=1e-7)
    np.testing.assert_allclose(b.cpu().numpy(), g[tag + 'fused_boxes'], rtol=0, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('conf_type,overflow,weights', [('avg', False, None), ('max', False, None), ('avg', True, None),
                                                        ('avg', False, [1.0, 0.5, 2.0, 1.0, 1.5])])
def test_fusion_batched_vs_oracle(device, conf_type, overflow, weights):
    """Several frames in one call, the other confidence rules and model weights, against the oracle per frame."""
    from detzero_amd import tta
    ops = ['tta_original', 'tta_flip_x', 'tta_rot_0.78539816', 'tta_scale_0.95', 'tta_flip_xy']
    frames = []
    m = 0
    for seed, n_obj in ((21, 40), (22, 5), (23, 90)):
        preds = gen.synth_predictions(seed, n_obj, ops)
        frames.append(preds)
        m = max(m, max(len(p['pred_boxes']) for p in preds))
    t = len(ops)
    pb, ps, pl = np.zeros((3, t, m, 7), np.float32), np.zeros((3, t, m), np.float32), np.zeros((3, t, m), np.int32)
    for f, preds in enumerate(frames):
        for i, p in enumerate(preds):
            n = len(p['pred_boxes'])
            pb[f, i, :n], ps[f, i, :n], pl[f, i, :n] = p['pred_boxes'].numpy(), p['pred_scores'].numpy(), p['pred_labels'].numpy()
    boxes = tta.restore_boxes(torch.from_numpy(pb).to(device), ops)
    ob, osc, ol, oc = tta.wbf_fuse_nosync(boxes.reshape(3, t * m, 7), torch.from_numpy(ps).to(device).reshape(3, t * m),
                                          torch.from_numpy(pl).to(device).reshape(3, t * m), t, weights=weights, conf_type=conf_type,
                                          allows_overflow=overflow)
    host = boxes.cpu().numpy()
    for f in range(3):
        rb, rs, rl = oracle_wbf.weighted_boxes_fusion_3d(host[f], ps[f], pl[f], weights=weights, conf_type=conf_type, allows_overflow=overflow)
        k = int(oc[f])
        assert k == rb.shape[0]
        np.testing.assert_array_equal(ol[f, :k].cpu().numpy(), rl)
        np.testing.assert_allclose(osc[f, :k].cpu().numpy(), rs, rtol=0, atol=1e-7)
        np.testing.assert_allclose(ob[f, :k].cpu().numpy(), rb, rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_fusion_edge_cases(device):
    from detzero_amd import tta
    z = torch.zeros((2, 6, 7), device=device)
    ob, osc, ol, oc = tta.wbf_fuse_nosync(z, torch.zeros((2, 6), device=device), torch.zeros((2, 6), dtype=torch.int32, device=device), 3)
    assert oc.tolist() == [0, 0]                                         # only padding rows
    box = torch.tensor([1.0, 2.0, 0.5, 4.0, 2.0, 1.5, 0.3], device=device)
    b = box.repeat(1, 4, 1).contiguous()                                # the same vehicle from 4 models, one below the score gate
    s = torch.tensor([[0.9, 0.8, 0.05, 0.7]], device=device)
    la = torch.ones((1, 4), dtype=torch.int32, device=device)
    ob, osc, ol, oc = tta.wbf_fuse_nosync(b, s, la, 4)
    assert oc.tolist() == [1] and ol[0, 0].item() == 1
    np.testing.assert_allclose(ob[0, 0].cpu().numpy(), box.cpu().numpy().astype(np.float64), atol=1e-6)
    np.testing.assert_allclose(osc[0, 0].item(), np.float32(np.float32((0.9 + 0.8 + 0.7) / 3) * np.float32(3)) / 4.0, atol=1e-7)


@pytest.mark.gpu
def test_tta_pipeline_end_to_end(device):
    """Detector over the copies of two small frames + restore + fusion in one sync-free call, against the same detector
    outputs pushed through the oracle's restore and fusion."""
    from detzero_amd import tta
    from detzero_amd.centerpoint import FramePipeline
    from detzero_amd.synth import VOXEL_SIZE_02
    from tests.util import make_model, masked_frame
    model, cfg, info = make_model(VOXEL_SIZE_02, seed=0)
    model = model.to(device)
    pipe = FramePipeline(model, info, math='f16x2')
    frames = [torch.from_numpy(masked_frame(31, 12000)[:9000].copy()).to(device), torch.from_numpy(masked_frame(32, 12000)[:9000].copy()).to(device)]
    aug = tta.TestTimeAugmentor([{'NAME': 'world_flip', 'ALONG_AXIS_LIST': ['x', 'xy']}, {'NAME': 'world_rotation', 'ROT_ANGLE': [0, 0.39365818]},
                                 {'NAME': 'world_scaling', 'SCALE_RANGE': [1.05]}])
    t = len(aug.op_names)
    assert t == 5
    ob, osc, ol, oc = tta.TTAPipeline(pipe, aug)(frames)
    # the same batch through the detector alone (same batch composition -> same kernels -> same boxes)
    out, counts = pipe([c for f in frames for c in aug.augment(f)])
    out, counts = out.cpu().numpy().reshape(2, t, -1, 9), counts.cpu().numpy().reshape(2, t)
    k = out.shape[2]
    for f in range(2):
        pb = out[f, :, :, :7].astype(np.float32)
        pl = np.where(np.arange(k)[None, :] < counts[f][:, None], out[f, :, :, 8], 0).astype(np.int32)
        rb, rs, rl = oracle_wbf.weighted_boxes_fusion_3d(oracle_wbf.restore_boxes(pb, aug.op_names), out[f, :, :, 7], pl)
        n = int(oc[f])
        assert n == rb.shape[0] and n > 0
        # a random-init detector emits many boxes with identical scores (constant background response); the order among equal
        # scores is unspecified in the reference (numpy argsort), so both results are put into one canonical order first
        mine = np.concatenate([osc[f, :n].cpu().numpy()[:, None], ol[f, :n].cpu().numpy()[:, None], ob[f, :n].cpu().numpy()], axis=1)
        ref = np.concatenate([rs[:, None], rl[:, None], rb], axis=1)
        mine = mine[np.lexsort(np.round(mine[:, ::-1], 3).T)]
        ref = ref[np.lexsort(np.round(ref[:, ::-1], 3).T)]
        np.testing.assert_allclose(mine[:, :2], ref[:, :2], rtol=0, atol=1e-6)
        np.testing.assert_allclose(mine[:, 2:8], ref[:, 2:8], rtol=0, atol=1e-4)
        # heading = the heading of the cluster's most confident member: with equal confidences (see above) either may be it
        assert (np.abs(mine[:, 8] - ref[:, 8]) > 1e-4).mean() < 0.02


# ------------------------------------------------------------------------------------------------ tracking variant (object ids)
def test_oracle_tracking_fusion_equals_reference(g):
    tag = 'f%d_' % gen.FRAMES[1][0]
    b, s, l, ids = oracle_wbf.weighted_boxes_fusion_3d(g[tag + 'restored'], g[tag + 'pred_scores'], g[tag + 'pred_labels'], obj_ids=g[tag + 'obj_ids'])
    np.testing.assert_array_equal(b, g[tag + 'trk_boxes'])
    np.testing.assert_array_equal(s, g[tag + 'trk_scores'])
    np.testing.assert_array_equal(l, g[tag + 'trk_labels'])
    np.testing.assert_array_equal(ids, g[tag + 'trk_ids'])
    assert (ids >= 0).sum() > 10 and (ids < 0).sum() > 0


@pytest.mark.gpu
def test_tracking_fusion_matches_reference(device, g):
    """wbf_tracking_v1 (ensemble.py:35-62 -> weighted_tracking_boxes_fusion_3d): the fused boxes of wbf_online plus, per box, the
    object id of its most confident member that has one."""
    from detzero_amd import tta
    tag = 'f%d_' % gen.FRAMES[1][0]
    b, s, l, ids = tta.wbf_tracking_v1(torch.from_numpy(g[tag + 'restored']).to(device), torch.from_numpy(g[tag + 'pred_scores']).to(device)[..., None],
                                       torch.from_numpy(g[tag + 'pred_labels']).to(device)[..., None], torch.from_numpy(g[tag + 'obj_ids']).to(device)[..., None])
    assert ids.dtype == torch.int64 and b.shape[0] == g[tag + 'trk_boxes'].shape[0]
    np.testing.assert_array_equal(l.cpu().numpy(), g[tag + 'trk_labels'])
    np.testing.assert_array_equal(ids.cpu().numpy(), g[tag + 'trk_ids'])
    np.testing.assert_allclose(s.cpu().numpy(), g[tag + 'trk_scores'], rtol=0, atol=1e-7)
    np.testing.assert_allclose(b.cpu().numpy(), g[tag + 'trk_boxes'], rtol=0, atol=1e-6)


def _tie_case():
    """Two models, three objects; members of a cluster with EQUAL confidences and different object ids."""
    base = np.array([[10, 5, 1, 4, 2, 1.6, 0.3], [-20, 8, 1, 4.2, 1.9, 1.5, -1.0], [30, -12, 1, 0.8, 0.8, 1.7, 2.0]], np.float32)
    boxes = np.stack([base, base + np.array([0.05, 0.02, 0, 0, 0, 0, 0.01], np.float32)])
    scores = np.array([[0.8, 0.6, 0.5], [0.8, 0.6, 0.4]], np.float32)
    labels = np.array([[1, 1, 2], [1, 1, 2]], np.int64)
    ids = np.array([[5, -1, 9], [7, 3, 11]], np.int64)
    return boxes, scores, labels, ids


def test_oracle_tracking_tie_rule():
    """get_weighted_box orders members by np.argsort(conf)[::-1] (wbf_3d.py:86-94): among equal confidences the member
    appended LAST lends its id."""
    boxes, scores, labels, ids = _tie_case()
    b, s, l, out = oracle_wbf.weighted_boxes_fusion_3d(boxes, scores, labels, obj_ids=ids)
    assert b.shape[0] == 3
    order = np.argsort(-s)
    assert out[order].tolist() == [5, 3, 9]


@pytest.mark.gpu
def test_tracking_fusion_tie_rule_on_device(device):
    from detzero_amd import tta
    boxes, scores, labels, ids = _tie_case()
    rb, rs, rl, rid = oracle_wbf.weighted_boxes_fusion_3d(boxes, scores, labels, obj_ids=ids)
    b, s, l, out = tta.wbf_tracking_v1(torch.from_numpy(boxes).to(device), torch.from_numpy(scores).to(device)[..., None],
                                       torch.from_numpy(labels).to(device)[..., None], torch.from_numpy(ids).to(device)[..., None])
    np.testing.assert_array_equal(out.cpu().numpy(), rid)
    np.testing.assert_allclose(s.cpu().numpy(), rs, rtol=0, atol=1e-7)
    np.testing.assert_allclose(b.cpu().numpy()[:, :6], rb[:, :6], rtol=0, atol=1e-6)


def test_wbf_weights_length_mismatch_falls_back_to_ones_like_the_reference():
    """wbf_3d.py:143-145 resets a weights list of the wrong length to ones; the host wrapper must not hand it to the kernel."""
    import inspect
    from detzero_amd import tta
    src = inspect.getsource(tta.wbf_fuse_nosync)
    assert 'len(weights) != n_models' in src and 'weights = None' in src


def test_augmentor_config_forms_and_errors():
    from detzero_amd.config import AttrDict
    from detzero_amd.lib import DetZeroHipError
    from detzero_amd.tta import TestTimeAugmentor, parse_op
    cfg = AttrDict({'DISABLE_AUG_LIST': ['world_scaling'], 'AUG_CONFIG_LIST': [AttrDict(c) for c in gen.AUG_CONFIG]})
    aug = TestTimeAugmentor(cfg)
    assert len(aug.op_names) == 13 and not any('scale' in n for n in aug.op_names)
    assert aug.kinds.dtype == np.int32 and aug.params.dtype == np.float32 and aug.kinds[0] == 0
    for bad in ('tta_shear_1', 'flip_x', 'tta_flip_z', 'tta_rot'):
        with pytest.raises((DetZeroHipError, ValueError)):
            parse_op(bad)
    with pytest.raises(DetZeroHipError):
        TestTimeAugmentor([{'NAME': 'world_translation'}])
