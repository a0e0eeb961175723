"""Import shims: the package names the reference's tools import, re-exporting the MI355X mirrors.

``detection/tools/test.py`` and ``eval_utils.py`` of the reference import (test.py:1-19, eval_utils.py:1-11)

    tensorboardX.SummaryWriter, detzero_utils.{common_utils, config_utils, model_utils},
    detzero_det.datasets.build_dataloader, detzero_det.models.{build_network, load_data_to_gpu}

Putting THIS directory on ``PYTHONPATH`` (``python -m detzero_amd.shim`` prints it) makes those names resolve to
``detzero_amd`` - the reference's ``test.py`` then runs unchanged on the HIP backend (``--workers 0``: frames are
voxelized on the device, which forked DataLoader workers cannot own).  ``tensorboardX`` / ``easydict`` fall-backs live in
``_fallbacks`` and are appended at the END of ``sys.path`` by ``install()``, so real installations win.
"""
import os
import sys

SHIM_DIR = os.path.dirname(os.path.abspath(__file__))
FALLBACK_DIR = os.path.join(SHIM_DIR, '_fallbacks')


def install():
    """Make ``detzero_det`` / ``detzero_utils`` (and, if missing, ``tensorboardX`` / ``easydict``) importable in this process."""
    root = os.path.dirname(os.path.dirname(SHIM_DIR))
    for p in (root, SHIM_DIR):
        if p not in sys.path:
            sys.path.insert(0, p)
    if FALLBACK_DIR not in sys.path:
        sys.path.append(FALLBACK_DIR)
    return SHIM_DIR
