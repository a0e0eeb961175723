"""detzero_utils.config_utils - the global ``cfg`` and the three functions the reference's tools call (test.py:1-19):
cfg_from_yaml_file (with ``_BASE_CONFIG_`` includes, resolved against the working directory because the reference's tools run
from detection/tools), cfg_from_list (``KEY.SUB value`` overrides from the command line) and log_config_to_file."""
import ast

import yaml

from detzero_amd.config import AttrDict

EasyDict = AttrDict


def _walk(node, prefix):
    """(dotted path, value, is a section) for every entry, depth first in insertion order."""
    for key, val in node.items():
        path = '%s.%s' % (prefix, key)
        section = isinstance(val, AttrDict)
        yield path, val, section
        if section:
            yield from _walk(val, path)


def log_config_to_file(cfg, pre='cfg', logger=None):
    for path, val, section in _walk(cfg, pre):
        logger.info('\n%s = edict()' % path if section else '%s: %s' % (path, val))


def _parse(text):
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return text


def _override(section, key, text):
    """One command-line override: the new value takes the type of the entry it replaces."""
    old, new = section[key], _parse(text)
    if isinstance(old, AttrDict) and not isinstance(new, dict):               # "a:1,b:2" into a sub-section
        for item in str(text).split(','):
            k, v = item.split(':')
            old[k] = type(old[k])(v)
    elif isinstance(old, list) and not isinstance(new, list):                 # "1,2,3" into a list
        section[key] = [type(old[0])(x) for x in str(text).split(',')]
    elif type(new) is type(old):
        section[key] = new
    else:
        raise AssertionError('type %s does not match original type %s' % (type(new), type(old)))


def cfg_from_list(cfg_list, config):
    if len(cfg_list) % 2:
        raise AssertionError('overrides come in KEY VALUE pairs')
    for dotted, text in zip(cfg_list[0::2], cfg_list[1::2]):
        *parents, leaf = dotted.split('.')
        section = config
        for name in parents:
            assert name in section, 'NotFoundKey: %s' % name
            section = section[name]
        assert leaf in section, 'NotFoundKey: %s' % leaf
        _override(section, leaf, text)


def merge_new_config(config, new_config):
    """Deep-merge a loaded yaml dict into `config`; a ``_BASE_CONFIG_`` entry is loaded first, underneath."""
    base = new_config.get('_BASE_CONFIG_')
    if base is not None:
        with open(base) as f:
            config.update(AttrDict(yaml.safe_load(f)))
    for key, val in new_config.items():
        if isinstance(val, dict):
            merge_new_config(config.setdefault(key, AttrDict()), val)
        else:
            config[key] = val
    return config


def cfg_from_yaml_file(cfg_file, config):
    with open(cfg_file) as f:
        return merge_new_config(config, yaml.safe_load(f))


cfg = AttrDict()
cfg.LOCAL_RANK = 0
