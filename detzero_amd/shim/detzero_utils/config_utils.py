"""detzero_utils.config_utils (utils/detzero_utils/config_utils.py:1-97): the global ``cfg``, yaml loading with
``_BASE_CONFIG_`` includes resolved against the CWD like the reference (its tools run from detection/tools),
``cfg_from_list`` command-line overrides, ``log_config_to_file``."""
from ast import literal_eval

import yaml

from detzero_amd.config import AttrDict

EasyDict = AttrDict


def log_config_to_file(cfg, pre='cfg', logger=None):
    """config_utils.py:6-12."""
    for key, val in cfg.items():
        if isinstance(cfg[key], AttrDict):
            logger.info('\n%s.%s = edict()' % (pre, key))
            log_config_to_file(cfg[key], pre=pre + '.' + key, logger=logger)
            continue
        logger.info('%s.%s: %s' % (pre, key, val))


def cfg_from_list(cfg_list, config):
    """config_utils.py:24-56: ``KEY.SUBKEY value`` pairs; dict-valued and list-valued keys take ``a:1,b:2`` / ``1,2`` strings."""
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        key_list = k.split('.')
        d = config
        for subkey in key_list[:-1]:
            assert subkey in d, 'NotFoundKey: %s' % subkey
            d = d[subkey]
        subkey = key_list[-1]
        assert subkey in d, 'NotFoundKey: %s' % subkey
        try:
            value = literal_eval(v)
        except Exception:
            value = v
        if type(value) != type(d[subkey]) and isinstance(d[subkey], AttrDict):
            for src in value.split(','):
                cur_key, cur_val = src.split(':')
                d[subkey][cur_key] = type(d[subkey][cur_key])(cur_val)
        elif type(value) != type(d[subkey]) and isinstance(d[subkey], list):
            d[subkey] = [type(d[subkey][0])(x) for x in value.split(',')]
        else:
            assert type(value) == type(d[subkey]), 'type {} does not match original type {}'.format(type(value), type(d[subkey]))
            d[subkey] = value


def merge_new_config(config, new_config):
    """config_utils.py:59-76 (the include path is opened relative to the CWD, as in the reference)."""
    if '_BASE_CONFIG_' in new_config:
        with open(new_config['_BASE_CONFIG_'], 'r') as f:
            config.update(AttrDict(yaml.safe_load(f)))
    for key, val in new_config.items():
        if not isinstance(val, dict):
            config[key] = val
            continue
        if key not in config:
            config[key] = AttrDict()
        merge_new_config(config[key], val)
    return config


def cfg_from_yaml_file(cfg_file, config):
    with open(cfg_file, 'r') as f:
        merge_new_config(config=config, new_config=yaml.safe_load(f))
    return config


cfg = AttrDict()
cfg.LOCAL_RANK = 0
