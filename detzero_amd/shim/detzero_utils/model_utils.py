"""detzero_utils.model_utils: checkpoint loading as the detection tools call it (utils/detzero_utils/model_utils.py:8-78).

One deliberate difference: the reference skips missing / mis-shaped keys SILENTLY, so a checkpoint with
the other sparse-conv weight layout "loads" and leaves random weights.  Here a 5-D sparse-conv weight whose shape is the
spconv-1.x / "Native" layout (kD,kH,kW,Cin,Cout) is converted to the (Cout,kD,kH,kW,Cin) layout the modules hold, and a
checkpoint that updates NO backbone3d weight at all raises."""
import os

import torch

from detzero_amd.lib import DetZeroHipError


def _convert_sparse_layout(key, val, want_shape):
    if val.dim() == 5 and tuple(val.shape) != tuple(want_shape) and tuple(val.permute(4, 0, 1, 2, 3).shape) == tuple(want_shape):
        return val.permute(4, 0, 1, 2, 3).contiguous()
    return val


def load_params_from_file(model, filename, logger, to_cpu=False, fix_pretrained_weights=False):
    if not os.path.isfile(filename):
        raise FileNotFoundError
    logger.info('==> Loading parameters from checkpoint %s to %s' % (filename, 'CPU' if to_cpu else 'GPU'))
    checkpoint = torch.load(filename, map_location=torch.device('cpu') if to_cpu else None, weights_only=False)
    model_state_disk = checkpoint['model_state']
    if 'version' in checkpoint:
        logger.info('==> Checkpoint trained from version: %s' % checkpoint['version'])
    state_dict = model.state_dict()
    update_model_state = {}
    for key, val in model_state_disk.items():
        if key not in state_dict:
            continue
        val = _convert_sparse_layout(key, val, state_dict[key].shape)
        if state_dict[key].shape == val.shape:
            update_model_state[key] = val
    if any(k.startswith('backbone3d.') for k in state_dict) and not any(k.startswith('backbone3d.') for k in update_model_state):
        raise DetZeroHipError('load_params_from_file: no backbone3d weight of %s matches the model (wrong sparse-conv layout?)' % filename)
    state_dict.update(update_model_state)
    model.load_state_dict(state_dict)
    if fix_pretrained_weights:
        for name, param in model.named_parameters():
            if name in update_model_state:
                param.requires_grad = False
    for key in state_dict:
        if key not in update_model_state:
            logger.info('Not updated weight %s: %s' % (key, str(state_dict[key].shape)))
    logger.info('==> Done (loaded %d/%d)' % (len(update_model_state), len(model.state_dict())))


def load_params_with_optimizer(model, filename, to_cpu=False, optimizer=None, logger=None):
    """model_utils.py:46-78 (the optimizer branch is kept for signature parity; training is out of scope)."""
    if not os.path.isfile(filename):
        raise FileNotFoundError
    logger.info('==> Loading parameters from checkpoint %s to %s' % (filename, 'CPU' if to_cpu else 'GPU'))
    checkpoint = torch.load(filename, map_location=torch.device('cpu') if to_cpu else None, weights_only=False)
    model.load_state_dict(checkpoint['model_state'])
    if optimizer is not None and checkpoint.get('optimizer_state') is not None:
        optimizer.load_state_dict(checkpoint['optimizer_state'])
    if 'version' in checkpoint:
        logger.info('==> Checkpoint trained from version: %s' % checkpoint['version'])
    logger.info('==> Done')
    return checkpoint.get('it', 0.0), checkpoint.get('epoch', -1)
