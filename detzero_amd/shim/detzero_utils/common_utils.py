"""detzero_utils.common_utils - the names the reference's detection tools import (test.py:1-19, eval_utils.py:1-11:
create_logger, set_random_seed, init_dist_pytorch, init_dist_slurm, get_dist_info, merge_results_dist, mask_points_by_range),
implemented for this backend: one process per GPU, backend "nccl" = RCCL on ROCm, and the per-rank result lists merged
with a collective instead of the reference's pickle files + barriers (utils/detzero_utils/common_utils.py:119-140)."""
import logging
import os
import random
import subprocess

import numpy as np
import torch
import torch.distributed as dist

from detzero_amd.data_processor import mask_points_by_range  # noqa: F401  (same name, same semantics: inclusive xy bounds)
from detzero_amd.frame_parallel import interleave_parts

_LOG_FORMAT = '%(asctime)s  %(levelname)5s  %(message)s'


def create_logger(log_file=None, rank=0, log_level=logging.INFO):
    """A console (+ optional file) logger; every rank but 0 only reports errors."""
    level = log_level if rank == 0 else logging.ERROR
    handlers = [logging.StreamHandler()] + ([logging.FileHandler(log_file)] if log_file else [])
    logger = logging.getLogger('detzero_utils.common_utils')
    logger.setLevel(level)
    for h in handlers:
        h.setLevel(level)
        h.setFormatter(logging.Formatter(_LOG_FORMAT))
        logger.addHandler(h)
    return logger


def set_random_seed(seed):
    """Seeds Python, numpy and torch (host and every visible GPU).  The reference also flips cuDNN's determinism switches; the
    kernels of this backend have no such mode (their results do not depend on run-to-run scheduling)."""
    os.environ['PYTHONHASHSEED'] = str(seed)
    for fn in (random.seed, np.random.seed, torch.manual_seed):
        fn(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def _bind_gpu(index):
    n = torch.cuda.device_count()
    if n:
        torch.cuda.set_device(index % n)
    return n


def init_dist_pytorch(tcp_port, local_rank, backend='nccl'):
    """Launcher-provided environment (torch.distributed.run exports RANK / WORLD_SIZE / MASTER_*): bind the process to its GPU and
    join the group.  Returns (GPUs on this node, rank) - the pair the reference's tools unpack."""
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: what RCCL needs on this driver stack
    n = _bind_gpu(local_rank)
    dist.init_process_group(backend=backend)
    return n, dist.get_rank()


def init_dist_slurm(tcp_port, local_rank, backend='nccl'):
    """SLURM launch (srun, one task per GPU): rank / world size from SLURM_PROCID / SLURM_NTASKS, the first host of the node list as
    the rendezvous address.  Returns (world size, rank)."""
    env = os.environ
    rank, world = int(env['SLURM_PROCID']), int(env['SLURM_NTASKS'])
    _bind_gpu(rank)
    hosts = subprocess.run(['scontrol', 'show', 'hostname', env['SLURM_NODELIST']], capture_output=True, text=True, check=True).stdout.split()
    env.update({'MASTER_ADDR': hosts[0], 'MASTER_PORT': str(tcp_port), 'WORLD_SIZE': str(world), 'RANK': str(rank)})
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group(backend=backend)
    return dist.get_world_size(), dist.get_rank()


def get_dist_info():
    """(rank, world size); (0, 1) outside a process group."""
    ready = dist.is_available() and dist.is_initialized()
    return (dist.get_rank(), dist.get_world_size()) if ready else (0, 1)


def merge_results_dist(result_part, size, tmpdir):
    """Rank 0 receives every rank's list of per-frame records and puts them back into dataset order (the strided sampler's
    inverse: zip the parts, cut the wrap-around padding); other ranks get None.  One object all-gather - `tmpdir` is accepted for
    the reference's call signature and not used (no files, no second barrier).  Box payloads on the hot path travel through
    detzero_amd.frame_parallel.gather_frame_boxes instead."""
    rank, world = get_dist_info()
    if world == 1:
        return list(result_part)[:size]
    parts = [None] * world
    dist.all_gather_object(parts, result_part)
    return interleave_parts(parts, size) if rank == 0 else None
