"""Golden vectors for the confidence refining model (CRM) from the REFERENCE's own classes, CPU.

    python tests/golden/gen_crm_golden.py      (build container only; needs /root/reference)

(1) ConfidencePointnet (refining/detzero_refine/models/modules/confidence_pointnet.py) in eval mode with the values of
ref_model_cfgs/cyclist_crm_model.yaml on seeded inputs; weights are the deterministic synth_state_dict(key, shape) as in
gen_refine_golden.py, the fixture keeps the key/shape manifest, inputs and outputs.
(2) WaymoConfidenceDataset.extract_track_feature (datasets/waymo/waymo_confidence_dataset.py:59-162) in inference mode
on seeded synthetic tracks + DatasetTemplate.collate_batch, constructor bypassed as in gen_refine_feat_golden.py.
"""
import importlib
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

CRM_TRACKS = [(51, 25, 'Cyclist', 0, 500), (52, 4, 'Cyclist', 5, 30), (53, 120, 'Vehicle', 0, 300)]
CCFG = {'ENCODER_MLP': [128, 128], 'REGRESSION_MLP': [512], 'SCORE_THRESH': [0.35, 0.7]}


def main():
    import gen_refine_feat_golden as feat
    from detzero_amd.config import AttrDict
    from detzero_amd.synth import synth_object_track, synth_state_dict
    out = {}
    # ---- (2) first: the dataset import sets up the package stubs (real common_utils, box_utils)
    Geo, Pos, Base = feat.import_reference()
    conf = importlib.import_module('detzero_refine.datasets.waymo.waymo_confidence_dataset')
    items = []
    for seed, n, name, lo, hi in CRM_TRACKS:
        ds = feat.bare(conf.WaymoConfidenceDataset, query_num=200, query_pts_num=256, encoding=['xyz', 'intensity', 'p2co', 'score'])
        random.seed(3000 + seed)
        info = feat.data_info(synth_object_track(seed, n, name, lo, hi))
        info['refine_iou'] = np.zeros(n)
        items.append(ds.extract_track_feature(info))
    batch = Base.collate_batch(items)
    out['feat_conf_points'] = np.asarray(batch['conf_points'], dtype=np.float32)
    out['feat_conf_score'] = np.asarray(batch['conf_score'], dtype=np.float64)
    out['feat_box_num'] = np.asarray(batch['box_num'])
    # ---- (1) the model
    import types
    sys.modules.setdefault('detzero_refine.models', types.ModuleType('detzero_refine.models')).__path__ = [feat.REF + '/refining/detzero_refine/models']
    sys.modules.setdefault('detzero_refine.models.modules', types.ModuleType('detzero_refine.models.modules')).__path__ = [feat.REF + '/refining/detzero_refine/models/modules']
    importlib.import_module('detzero_utils.model_utils')
    mod = importlib.import_module('detzero_refine.models.modules.confidence_pointnet')
    crm = mod.ConfidencePointnet(AttrDict(CCFG), query_point_dims=32, memory_point_dims=32).eval()
    sd = synth_state_dict({k: tuple(v.shape) for k, v in crm.state_dict().items()}, seed=7)
    crm.load_state_dict(sd, strict=True)
    gen = torch.Generator().manual_seed(91)
    b, nb, npts = 3, 200, 24
    pts = torch.randn((b, nb, npts, 32), generator=gen)
    for i, n in enumerate([200, 57, 1]):
        pts[i, n:] = 0
    with torch.no_grad():
        res = crm({'conf_points': pts.clone()})
    out['crm_keys'] = np.array(list(crm.state_dict().keys()))
    out['crm_shapes'] = np.array([str(tuple(v.shape)) for v in crm.state_dict().values()])
    out['crm_in_conf_points'] = pts.numpy()
    out['crm_score_reg'] = crm.preds_dict['score_reg'].numpy()
    out['crm_iou_reg'] = crm.preds_dict['iou_reg'].numpy()
    out['crm_pred_score'] = res['pred_score'].numpy()
    np.savez_compressed(os.path.join(HERE, 'crm_golden.npz'), **out)
    print('CRM', out['crm_pred_score'].shape, len(out['crm_keys']), 'state-dict entries; features', out['feat_conf_points'].shape,
          os.path.getsize(os.path.join(HERE, 'crm_golden.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
