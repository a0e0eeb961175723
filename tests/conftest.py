import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'experimental: exercises an engine that is only compiled with DZ_BUILD_EXPERIMENTAL=1 (the tile-resident '
                                       'sparse convolution, the direct-to-LDS gather variant); deselected unless that variable is set')
    import torch
    torch.set_num_threads(_usable_cores())      # the CPU oracle must not oversubscribe a quota-limited container


def pytest_collection_modifyitems(config, items):
    """Tests of the experimental engines are DESELECTED (not skipped) in a default build: `pytest -m gpu` reports what ships."""
    if os.environ.get('DZ_BUILD_EXPERIMENTAL', '0') not in ('', '0'):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker('experimental') else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda', 0)
