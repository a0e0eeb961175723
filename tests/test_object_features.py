"""Object features (cropped points -> GRM / PRM inputs): the oracle against the reference's own dataset classes (CPU),
the device path against both (GPU)."""
import os
import random
import sys

import numpy as np
import pytest
import torch

from detzero_amd.synth import synth_object_track, synth_state_dict
from oracle import object_features as oracle_feat

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import gen_refine_feat_golden as gen          # noqa: E402  (track lists and seeds of the fixture; importing it does not touch the reference)

GRM_KEYS = ('geo_query_points', 'geo_query_boxes', 'geo_memory_points')
PRM_KEYS = ('pos_trajectory', 'pos_init_box', 'padding_mask', 'pos_query_points', 'pos_memory_points')
PRM_ENC = ('xyz', 'intensity', 'p2co', 'score')


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'refine_feat_golden.npz'))


def _tracks(spec):
    return [synth_object_track(seed, n, name, lo, hi) for seed, n, name, lo, hi in spec]


# ------------------------------------------------------------------------------------------------ CPU: oracle vs reference
def test_oracle_grm_equals_reference(g):
    """Same seeds as the fixture -> the reference's numbers to the last bit (float64), including WHICH points were drawn."""
    objs = []
    for (seed, *_), tr in zip(gen.GRM_TRACKS, _tracks(gen.GRM_TRACKS)):
        random.seed(1000 + seed)
        objs.append(oracle_feat.grm_object(tr))
    batch = oracle_feat.grm_batch(objs)
    assert batch['geo_query_num'] == g['grm_query_num'].tolist()
    for k in GRM_KEYS:
        np.testing.assert_array_equal(batch[k], g['grm_' + k])


@pytest.mark.parametrize('tag,spec,enc', [('prm', gen.PRM_TRACKS, PRM_ENC), ('prmc', gen.PRM_CLASS_TRACKS, PRM_ENC + ('class',))])
def test_oracle_prm_equals_reference(g, tag, spec, enc):
    objs = []
    for (seed, *_), tr in zip(spec, _tracks(spec)):
        random.seed(2000 + seed)
        objs.append(oracle_feat.prm_object(tr, encoding=enc))
    batch = oracle_feat.prm_batch(objs)
    for k in PRM_KEYS:
        ref = g[tag + '_' + k]
        if ref.dtype == np.float32:          # the two big arrays are stored as float32
            # bit-equal except the p2co channels of a few rows: the corners are float32 arithmetic (torch.cos / matmul in the
            # reference, numpy in the oracle), 1 ulp apart here and there
            np.testing.assert_allclose(batch[k].astype(np.float32), ref, rtol=0, atol=1e-6)
            assert (batch[k].astype(np.float32) != ref).mean() < 0.01
        else:
            np.testing.assert_array_equal(batch[k], ref)


def test_selection_is_the_reference_draw():
    """Host index lists: one random.sample per over-full set in the reference's order; short sets keep every row."""
    from detzero_amd import object_features as of
    tracks = _tracks(gen.GRM_TRACKS)
    packed = of.PackedTracks(tracks, device=torch.device('cpu'))
    rng = random.Random(5)
    mem_idx, query_box, query_idx, qnum, orders = of.grm_selection(packed, rng=rng)
    chk = random.Random(5)
    for i, tr in enumerate(tracks):
        n = sum(p.shape[0] for p in tr['pts'])
        want = oracle_feat.draw_subset(n, 4096, chk)
        np.testing.assert_array_equal(mem_idx[i, :len(want)], want)
        assert (mem_idx[i, len(want):] == -1).all()
        order = np.argsort(tr['score'])[::-1][:3]
        assert qnum[i] == len(order)
        for q, f in enumerate(order):
            want = oracle_feat.draw_subset(tr['pts'][f].shape[0], 256, chk)
            assert query_box[i, q] == packed.obj_box_start[i] + f
            np.testing.assert_array_equal(query_idx[i, q, :len(want)], want)
            assert (query_idx[i, q, len(want):] == -1).all()
        assert (query_box[i, qnum[i]:] == -1).all()


# ------------------------------------------------------------------------------------------------ GPU
def _close(a, ref, atol, rtol=2e-6):
    torch.testing.assert_close(a.detach().cpu().double(), torch.from_numpy(np.asarray(ref, dtype=np.float64)), atol=atol, rtol=rtol)


@pytest.mark.gpu
def test_grm_features_match_reference(device, g):
    """Per object with the fixture's seeds: the device rows are the reference's rows rounded to float32."""
    from detzero_amd import object_features as of
    for i, ((seed, *_), tr) in enumerate(zip(gen.GRM_TRACKS, _tracks(gen.GRM_TRACKS))):
        random.seed(1000 + seed)
        out = of.grm_features([tr], device=device)
        q = out['geo_query_num'][0]
        assert q == int(g['grm_query_num'][i])
        _close(out['geo_memory_points'][0], g['grm_geo_memory_points'][i], atol=2e-6)
        _close(out['geo_query_points'][0], g['grm_geo_query_points'][i][:q], atol=2e-6)
        _close(out['geo_query_boxes'][0], g['grm_geo_query_boxes'][i][:q], atol=1e-6)


@pytest.mark.gpu
def test_grm_features_batched_vs_oracle(device):
    """One call over a batch (single RNG stream, queries padded to the batch's largest count) against the oracle."""
    from detzero_amd import object_features as of
    spec = gen.GRM_TRACKS + [(15, 60, 'Vehicle', 300, 900), (16, 3, 'Pedestrian', 0, 0)]
    tracks = _tracks(spec)
    random.seed(99)
    ref = oracle_feat.grm_batch([oracle_feat.grm_object(t) for t in tracks])
    random.seed(99)
    out = of.grm_features(tracks, device=device)
    assert out['geo_query_num'] == ref['geo_query_num']
    for k in GRM_KEYS:
        _close(out[k], ref[k], atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('tag,spec,enc', [('prm', gen.PRM_TRACKS, PRM_ENC), ('prmc', gen.PRM_CLASS_TRACKS, PRM_ENC + ('class',))])
def test_prm_features_match_reference(device, g, tag, spec, enc):
    from detzero_amd import object_features as of
    for i, ((seed, *_), tr) in enumerate(zip(spec, _tracks(spec))):
        random.seed(2000 + seed)
        out = of.prm_features([tr], encoding=enc, device=device)
        _close(out['pos_init_box'][0], g[tag + '_pos_init_box'][i], atol=1e-9, rtol=1e-12)
        _close(out['padding_mask'][0], g[tag + '_padding_mask'][i], atol=0)
        # float32 rounding of coordinates up to ~100 m in the middle box's frame; corners carry float32 cos/sin
        _close(out['pos_trajectory'][0], g[tag + '_pos_trajectory'][i], atol=1e-5)
        _close(out['pos_query_points'][0], g[tag + '_pos_query_points'][i], atol=3e-5)
        _close(out['pos_memory_points'][0], g[tag + '_pos_memory_points'][i], atol=3e-5)


@pytest.mark.gpu
def test_prm_features_batched_vs_oracle(device):
    from detzero_amd import object_features as of
    spec = gen.PRM_TRACKS + [(24, 1, 'Vehicle', 700, 700), (25, 5, 'Pedestrian', 0, 0)]
    tracks = _tracks(spec)
    random.seed(7)
    ref = oracle_feat.prm_batch([oracle_feat.prm_object(t) for t in tracks])
    random.seed(7)
    out = of.prm_features(tracks, device=device)
    assert out['box_num'] == ref['box_num']
    for k in PRM_KEYS:
        _close(out[k], ref[k], atol=3e-5 if 'points' in k else 1e-5)


@pytest.mark.gpu
def test_features_feed_the_refining_models(device):
    """crop-format tracks -> device features -> GRM / PRM forward, against the same models fed the oracle's features."""
    from detzero_amd import object_features as of
    from test_refine import _models
    grm, prm = _models()
    grm.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in grm.state_dict().items()}, seed=5), strict=True)
    prm.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in prm.state_dict().items()}, seed=6), strict=True)
    grm, prm = grm.to(device), prm.to(device)
    tracks = _tracks([(41, 12, 'Vehicle', 50, 400), (42, 30, 'Vehicle', 0, 200), (43, 7, 'Cyclist', 20, 120)])
    random.seed(3)
    ref = oracle_feat.grm_batch([oracle_feat.grm_object(t) for t in tracks])
    random.seed(3)
    mine = of.grm_features(tracks, device=device)
    a = grm({k: (v.clone() if torch.is_tensor(v) else torch.tensor(v)) for k, v in mine.items() if k != 'batch_size'})['batch_box_preds']
    b = grm({'geo_memory_points': torch.from_numpy(ref['geo_memory_points']).to(device), 'geo_query_points': torch.from_numpy(ref['geo_query_points']).to(device),
             'geo_query_boxes': torch.from_numpy(ref['geo_query_boxes']).to(device), 'geo_query_num': torch.tensor(ref['geo_query_num'])})['batch_box_preds']
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)
    random.seed(4)
    ref = oracle_feat.prm_batch([oracle_feat.prm_object(t) for t in tracks])
    random.seed(4)
    mine = of.prm_features(tracks, device=device)
    a = prm({k: mine[k] for k in ('pos_query_points', 'pos_memory_points', 'pos_trajectory', 'padding_mask')})['batch_box_preds']
    b = prm({k: torch.from_numpy(ref[k]).to(device) for k in ('pos_query_points', 'pos_memory_points', 'pos_trajectory', 'padding_mask')})['batch_box_preds']
    valid = torch.from_numpy(ref['padding_mask']) == 0
    torch.testing.assert_close(a.cpu()[valid], b.cpu()[valid], rtol=1e-3, atol=1e-3)


# ------------------------------------------------------------------------------------------------ CRM
@pytest.fixture(scope='module')
def gc(golden_dir):
    return np.load(os.path.join(golden_dir, 'crm_golden.npz'))


def _crm_spec():
    import gen_crm_golden
    return gen_crm_golden.CRM_TRACKS, gen_crm_golden.CCFG


def test_oracle_crm_features_equal_reference(gc):
    spec, _ = _crm_spec()
    objs = []
    for (seed, *_), tr in zip(spec, _tracks(spec)):
        random.seed(3000 + seed)
        objs.append(oracle_feat.crm_object(tr))
    batch = oracle_feat.crm_batch(objs)
    assert batch['box_num'] == gc['feat_box_num'].tolist()
    np.testing.assert_array_equal(batch['conf_score'], gc['feat_conf_score'])
    np.testing.assert_allclose(batch['conf_points'].astype(np.float32), gc['feat_conf_points'], rtol=0, atol=1e-6)      # p2co corners: 1 ulp, see PRM


def test_crm_state_dict_identical_to_reference_manifest(gc):
    from detzero_amd.config import AttrDict
    from detzero_amd.refine_modules import ConfidencePointnet
    _, cfg = _crm_spec()
    crm = ConfidencePointnet(AttrDict(cfg), query_point_dims=32, memory_point_dims=32)
    mine = {k: str(tuple(v.shape)) for k, v in crm.state_dict().items()}
    assert list(mine) == gc['crm_keys'].tolist()
    assert list(mine.values()) == gc['crm_shapes'].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize('math', ['f32', 'f16x2'])
def test_crm_model_matches_reference(device, gc, math):
    from detzero_amd.config import AttrDict
    from detzero_amd.refine_modules import ConfidencePointnet
    _, cfg = _crm_spec()
    crm = ConfidencePointnet(AttrDict(cfg), query_point_dims=32, memory_point_dims=32).eval()
    crm.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in crm.state_dict().items()}, seed=7), strict=True)
    crm = crm.to(device).set_math(math)
    res = crm({'conf_points': torch.from_numpy(gc['crm_in_conf_points']).to(device)})
    torch.testing.assert_close(crm.preds_dict['score_reg'].cpu(), torch.from_numpy(gc['crm_score_reg']), rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(crm.preds_dict['iou_reg'].cpu(), torch.from_numpy(gc['crm_iou_reg']), rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(res['pred_score'].cpu(), torch.from_numpy(gc['crm_pred_score']), rtol=1e-3, atol=2e-4)


@pytest.mark.gpu
def test_crm_features_match_reference_and_feed_the_model(device, gc):
    from detzero_amd import object_features as of
    from detzero_amd.config import AttrDict
    from detzero_amd.refine_modules import ConfidencePointnet
    spec, cfg = _crm_spec()
    tracks = _tracks(spec)
    for i, ((seed, *_), tr) in enumerate(zip(spec, tracks)):
        random.seed(3000 + seed)
        out = of.crm_features([tr], device=device)
        assert out['box_num'] == [int(gc['feat_box_num'][i])]
        np.testing.assert_array_equal(out['conf_score'][0], gc['feat_conf_score'][i])
        _close(out['conf_points'][0], gc['feat_conf_points'][i], atol=3e-5)
    crm = ConfidencePointnet(AttrDict(cfg), query_point_dims=32, memory_point_dims=32).eval()
    crm.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in crm.state_dict().items()}, seed=7), strict=True)
    crm = crm.to(device)
    random.seed(8)
    feats = of.crm_features(tracks, device=device)
    a = crm({'conf_points': feats['conf_points']})['pred_score']
    b = crm({'conf_points': torch.from_numpy(gc['feat_conf_points']).to(device)})['pred_score']
    assert a.shape == (3, 200) and torch.isfinite(a).all()
    # different draws of the over-full boxes -> slightly different scores; same inputs elsewhere
    assert (a - b).abs().max() < 0.2


# ------------------------------------------------------------------------------------------------ device-side draw
def test_device_draw_restatement_is_uniform_and_sorted():
    n, k = 40, 10
    hits = np.zeros(n)
    for s in range(3000):
        idx = oracle_feat.device_draw_subset(n, k, 99, s)
        assert len(idx) == k and (np.diff(idx) > 0).all() and idx.min() >= 0 and idx.max() < n
        hits[idx] += 1
    assert abs(hits / 3000 - k / n).max() < 0.04                       # 5 sigma of a binomial(3000, 0.25) frequency
    np.testing.assert_array_equal(oracle_feat.device_draw_subset(7, 10, 1, 0), np.arange(7))


@pytest.mark.gpu
def test_device_draw_matches_restatement(device):
    from detzero_amd import object_features as of
    counts = [0, 5, 47, 48, 49, 255, 256, 257, 1000, 5000, 48, 48]
    for k, stream in ((48, 4), (256, 3)):
        dd = of.DeviceDraw(seed=20260925)
        got = dd.draw(counts, k, stream, device).cpu().numpy()
        seed = oracle_feat.device_stream_seed(20260925, stream)
        assert seed == dd.stream_seed(stream)
        for s, n in enumerate(counts):
            want = oracle_feat.device_draw_subset(n, k, seed, s)
            np.testing.assert_array_equal(got[s, :len(want)], want)
            assert (got[s, len(want):] == -1).all()


@pytest.mark.gpu
def test_features_with_device_draw_vs_oracle(device):
    """rng=DeviceDraw: nothing drawn on the host; the result equals the oracle fed the same (restated) draws."""
    from detzero_amd import object_features as of
    tracks = _tracks(gen.GRM_TRACKS + [(15, 30, 'Vehicle', 300, 900)])
    dd = of.DeviceDraw(seed=7)
    out = of.grm_features(tracks, rng=dd, device=device)
    q_max = max(out['geo_query_num'])
    objs = []
    for i, tr in enumerate(tracks):
        queue = []
        if sum(p.shape[0] for p in tr['pts']) >= 4096:
            queue.append((oracle_feat.device_stream_seed(7, 1), i))
        for q, f in enumerate(np.argsort(tr['score'])[::-1][:3]):
            if tr['pts'][f].shape[0] >= 256:
                queue.append((oracle_feat.device_stream_seed(7, 2), i * q_max + q))
        objs.append(oracle_feat.grm_object(tr, rng=oracle_feat.DeviceDrawReplay(queue)))
    ref = oracle_feat.grm_batch(objs)
    for k in GRM_KEYS:
        _close(out[k], ref[k], atol=2e-6)
    tracks = _tracks(gen.PRM_TRACKS[:2] + [(24, 6, 'Vehicle', 300, 700)])
    out = of.prm_features(tracks, rng=dd, device=device)
    objs, f = [], 0
    for tr in tracks:
        queue = []
        for p in tr['pts']:
            if p.shape[0] >= 256:
                queue.append((oracle_feat.device_stream_seed(7, 3), f))
            if p.shape[0] >= 48:
                queue.append((oracle_feat.device_stream_seed(7, 4), f))
            f += 1
        objs.append(oracle_feat.prm_object(tr, rng=oracle_feat.DeviceDrawReplay(queue)))
    ref = oracle_feat.prm_batch(objs)
    for k in PRM_KEYS:
        _close(out[k], ref[k], atol=3e-5 if 'points' in k else 1e-5)


def test_host_side_errors():
    from detzero_amd import object_features as of
    from detzero_amd.lib import DetZeroHipError
    cpu = torch.device('cpu')
    with pytest.raises(DetZeroHipError):
        of.PackedTracks([], device=cpu)
    tr = synth_object_track(1, 3, 'Vehicle', 1, 5)
    tr['score'] = tr['score'][:2]
    with pytest.raises(DetZeroHipError):
        of.PackedTracks([tr], device=cpu)
    with pytest.raises(DetZeroHipError):
        of.grm_features(of.PackedTracks([synth_object_track(2, 3, 'Vehicle', 1, 5)], device=cpu), encoding=('xyz', 'p2co'))
    with pytest.raises(DetZeroHipError):
        of.prm_features(of.PackedTracks([synth_object_track(3, 5, 'Vehicle', 1, 5)], device=cpu), query_num=4)
