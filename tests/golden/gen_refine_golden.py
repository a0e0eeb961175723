"""Golden vectors for the refining module (GRM / PRM) from the REFERENCE's own Python classes, CPU torch.

    python tests/golden/gen_refine_golden.py      (build container only; needs /root/reference)

Runs refining/detzero_refine/models/modules/{geometry_transformer.py, position_transformer.py} (with their
heads, decoder layer, attention, FFN, position embedding and target assigner) in eval mode with the
default vehicle configs (ref_model_cfgs/vehicle_{grm,prm}_model.yaml values) on small seeded inputs.
Weights are NOT stored: every state-dict entry is a deterministic function of (key name, shape)
(`detzero_amd.synth.synth_state_dict`), so the fixture holds the reference's key/shape manifest (which the
tests use to prove state-dict compatibility), the inputs and the outputs.
Parent package __init__ files (datasets, TF evaluators) are bypassed with path-only stub packages; nothing of
the arithmetic under test is stubbed.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def import_reference():
    torch.Tensor.cuda = lambda self, *a, **k: self
    _pkg('detzero_utils', REF + '/utils/detzero_utils')
    _pkg('detzero_refine', REF + '/refining/detzero_refine')
    _pkg('detzero_refine.utils', REF + '/refining/detzero_refine/utils')
    _pkg('detzero_refine.models', REF + '/refining/detzero_refine/models')
    _pkg('detzero_refine.models.modules', REF + '/refining/detzero_refine/models/modules')
    sys.modules['detzero_utils.common_utils'] = types.ModuleType('detzero_utils.common_utils')
    sys.modules['detzero_utils'].common_utils = sys.modules['detzero_utils.common_utils']
    importlib.import_module('detzero_utils.model_utils')
    geo = importlib.import_module('detzero_refine.models.modules.geometry_transformer')
    pos = importlib.import_module('detzero_refine.models.modules.position_transformer')
    return geo.GeometryTransformer, pos.PositionTransformer


def main():
    from detzero_amd.config import AttrDict
    from detzero_amd.synth import synth_state_dict
    Geo, Pos = import_reference()
    out = {}
    gen = torch.Generator().manual_seed(77)

    # ---------------- GRM (vehicle_grm_model.yaml), B=3 objects, 3 proposals x 64 query pts, 320 memory pts
    gcfg = AttrDict({'QUERY_ENCODER': [128, 128], 'MEMORY_ENCODER': [128, 128], 'REGRESSION_MLP': [512], 'EMBED_DIMS': 256,
                     'ANCHOR_SIZES': [[4.8, 1.8, 1.5], [10.0, 2.6, 3.2], [2.0, 1.0, 1.6]],
                     'DECODER': {'NAME': 'GeometryHead', 'num_classes': 3, 'num_heads': 8, 'num_decoder_layers': 1,
                                 'auxiliary': True, 'cross_only': False, 'memory_self_attn': False, 'hidden_channel': 256,
                                 'ffn_channel': 256, 'dropout': 0.1, 'bn_momentum': 0.1, 'activation': 'relu'}})
    grm = Geo(gcfg, query_point_dims=11, memory_point_dims=4).eval()
    sd = synth_state_dict({k: tuple(v.shape) for k, v in grm.state_dict().items()}, seed=5)
    grm.load_state_dict(sd, strict=True)
    b, lm = 3, 320
    data = {'geo_memory_points': torch.randn((b, lm, 11), generator=gen),
            'geo_query_points': torch.randn((b, 3, 64, 4), generator=gen),
            'geo_query_boxes': torch.randn((b, 3, 7), generator=gen),
            'geo_query_num': torch.tensor([3, 2, 1])}
    inp = {k: v.clone() for k, v in data.items()}
    with torch.no_grad():
        res = grm(data)
    out['grm_keys'] = np.array(list(grm.state_dict().keys()))
    out['grm_shapes'] = np.array([str(tuple(v.shape)) for v in grm.state_dict().values()])
    for k, v in inp.items():
        out['grm_in_' + k] = v.numpy()
    out['grm_query'] = res['query'].numpy()               # (B,256,3)
    out['grm_memory'] = res['memory'].numpy()             # (B,256,Lm)
    out['grm_cls'] = grm.preds_dict['geometry_cls'].numpy()
    out['grm_reg'] = grm.preds_dict['geometry_reg'].numpy()
    out['grm_boxes'] = res['batch_box_preds'].numpy()
    print('GRM', out['grm_cls'].shape, out['grm_reg'].shape, out['grm_boxes'].shape, len(out['grm_keys']), 'state-dict entries')

    # ---------------- PRM (vehicle_prm_model.yaml), B=2 tracks, 200 boxes x 16 query pts, 200 x 48 memory pts
    pcfg = AttrDict({'QUERY_ENCODER': [128, 128], 'MEMORY_ENCODER': [128, 128], 'REGRESSION_MLP': [512],
                     'LOSS_CLS': {'type': 'CrossEntropyLoss'},
                     'DECODER': {'NAME': 'PositionHead', 'num_classes': 3, 'num_heads': 8, 'num_decoder_layers': 1,
                                 'auxiliary': True, 'cross_only': False, 'hidden_channel': 256, 'dropout': 0.1,
                                 'bn_momentum': 0.1, 'activation': 'relu', 'ffn_channel': 256}})
    prm = Pos(pcfg, query_point_dims=32, memory_point_dims=32).eval()
    sd = synth_state_dict({k: tuple(v.shape) for k, v in prm.state_dict().items()}, seed=6)
    prm.load_state_dict(sd, strict=True)
    b, nb = 2, 200
    lens = [137, 200]
    pad = torch.zeros((b, nb))
    for i, n in enumerate(lens):
        pad[i, n:] = 1
    data = {'pos_query_points': torch.randn((b, nb, 16, 32), generator=gen),
            'pos_memory_points': torch.randn((b, nb, 48, 32), generator=gen),
            'pos_trajectory': torch.randn((b, nb, 7), generator=gen),
            'padding_mask': pad}
    inp = {k: v.clone() for k, v in data.items()}
    with torch.no_grad():
        res = prm(data)
    out['prm_keys'] = np.array(list(prm.state_dict().keys()))
    out['prm_shapes'] = np.array([str(tuple(v.shape)) for v in prm.state_dict().values()])
    for k, v in inp.items():
        out['prm_in_' + k] = v.numpy()
    out['prm_query'] = res['query'].numpy()
    out['prm_memory'] = res['memory'].numpy()[:, :, ::16].copy()       # subsampled (B,256,600) to keep the fixture small
    for k in ('center_reg', 'heading_cls', 'heading_reg'):
        out['prm_' + k] = prm.preds_dict[k].numpy()
    out['prm_boxes'] = res['batch_box_preds'].numpy()
    print('PRM', out['prm_center_reg'].shape, out['prm_boxes'].shape, len(out['prm_keys']), 'state-dict entries')
    np.savez_compressed(os.path.join(HERE, 'refine_golden.npz'), **out)
    print('saved', os.path.getsize(os.path.join(HERE, 'refine_golden.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
