"""Fall-back for easydict (absent from the ROCm image): the attribute-access dict of detzero_amd.config."""
from detzero_amd.config import AttrDict as EasyDict

__all__ = ['EasyDict']
