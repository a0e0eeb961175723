"""Refiner write-back (predictions -> per-frame boxes / result records) against the reference's own dataset classes."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import gen_refine_out_golden as gen          # noqa: E402  (the synthetic batch of the fixture; importing it does not touch the reference)


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'refine_out_golden.npz'))


def _check(tag, res, g):
    n = 0
    for seq in res:
        for obj, rec in res[seq].items():
            for k, v in rec.items():
                ref = g['%s_%s_%s_%s' % (tag, seq, obj, k)]
                if k in ('sequence_name', 'state', 'name'):
                    assert np.asarray(v).astype(str).tolist() == ref.tolist()
                else:
                    # float64 host arithmetic on coordinates of ~1e4 m: equal to the last few bits (the contraction order of the
                    # pose product may differ from the reference's einsum)
                    np.testing.assert_allclose(np.asarray(v, dtype=np.float64), ref.astype(np.float64), rtol=0, atol=1e-9,
                                               err_msg='%s %s %s' % (tag, obj, k))
                n += 1
    assert n == sum(1 for k in g.files if k.startswith(tag + '_'))


def test_write_back_equals_reference(g):
    from detzero_amd import refine_results as rr
    b, grm_pred, prm_pred, crm_pred = gen.synth_batch()
    _check('grm', rr.grm_prediction_dicts(b, grm_pred), g)
    _check('prm', rr.prm_prediction_dicts(b, prm_pred), g)
    _check('crm', rr.crm_prediction_dicts(b, crm_pred), g)


def test_round_trip_through_the_object_frame():
    """box_coords_transform undoes init_coords_transform (oracle), world_to_lidar with identity poses is the identity."""
    from detzero_amd import refine_results as rr
    from detzero_amd.synth import synth_object_track
    from oracle import object_features as of
    tr = synth_object_track(71, 9, 'Vehicle', 5, 20)
    obj = of.prm_object(tr)
    back = rr.box_coords_transform(obj['pos_trajectory'][:9].astype(np.float64), obj['pos_init_box'])
    want = tr['boxes_global'].copy()
    want[:, 6] = of.wrap_heading(want[:, 6])
    np.testing.assert_allclose(back[:, :6], want[:, :6], rtol=0, atol=2e-3)      # the yaw matrix is float32: ~1e-7 x 1e4 m
    np.testing.assert_allclose(np.cos(back[:, 6] - want[:, 6]), 1.0, atol=1e-9)
    eye = np.tile(np.eye(4), (9, 1, 1))
    np.testing.assert_allclose(rr.world_to_lidar(want, eye), want, rtol=0, atol=1e-9)
