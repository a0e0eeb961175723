"""detzero_utils.common_utils, the functions the detection tools call (utils/detzero_utils/common_utils.py:19-140,247-250):
logger, seeding, process-group set-up (backend "nccl" IS RCCL on ROCm), rank info and the rank-0 merge of per-rank results."""
import logging
import os
import pickle
import random
import shutil
import subprocess

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from detzero_amd.data_processor import mask_points_by_range  # noqa: F401  (common_utils.py:247-250)
from detzero_amd.frame_parallel import interleave_parts


def create_logger(log_file=None, rank=0, log_level=logging.INFO):
    """common_utils.py:19-33."""
    logger = logging.getLogger(__name__)
    logger.setLevel(log_level if rank == 0 else 'ERROR')
    formatter = logging.Formatter('%(asctime)s  %(levelname)5s  %(message)s')
    console = logging.StreamHandler()
    console.setLevel(log_level if rank == 0 else 'ERROR')
    console.setFormatter(formatter)
    logger.addHandler(console)
    if log_file is not None:
        file_handler = logging.FileHandler(filename=log_file)
        file_handler.setLevel(log_level if rank == 0 else 'ERROR')
        file_handler.setFormatter(formatter)
        logger.addHandler(file_handler)
    return logger


def set_random_seed(seed):
    """common_utils.py:51-60 (the cudnn switches have no MIOpen counterpart on this path: nothing here calls it)."""
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def init_dist_slurm(tcp_port, local_rank, backend='nccl'):
    """common_utils.py:63-84."""
    proc_id = int(os.environ['SLURM_PROCID'])
    ntasks = int(os.environ['SLURM_NTASKS'])
    node_list = os.environ['SLURM_NODELIST']
    num_gpus = torch.cuda.device_count()
    torch.cuda.set_device(proc_id % num_gpus)
    addr = subprocess.getoutput('scontrol show hostname {} | head -n1'.format(node_list))
    os.environ['MASTER_PORT'] = str(tcp_port)
    os.environ['MASTER_ADDR'] = addr
    os.environ['WORLD_SIZE'] = str(ntasks)
    os.environ['RANK'] = str(proc_id)
    dist.init_process_group(backend=backend)
    return dist.get_world_size(), dist.get_rank()


def init_dist_pytorch(tcp_port, local_rank, backend='nccl'):
    """common_utils.py:86-99: env:// rendezvous (torch.distributed.run exports RANK / WORLD_SIZE / MASTER_*), one process per GPU.
    Returns (GPUs of this node, rank) like the reference."""
    if mp.get_start_method(allow_none=True) is None:
        mp.set_start_method('spawn')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC for RCCL on this driver stack
    num_gpus = torch.cuda.device_count()
    if num_gpus:
        torch.cuda.set_device(local_rank % num_gpus)
    dist.init_process_group(backend=backend)
    return num_gpus, dist.get_rank()


def get_dist_info():
    """common_utils.py:102-116."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def merge_results_dist(result_part, size, tmpdir):
    """common_utils.py:119-140: per-rank pickle files + two barriers, rank 0 re-interleaves and truncates (the format
    eval_one_epoch expects; detzero_amd.frame_parallel.gather_frame_boxes is the collective replacement for box payloads)."""
    rank, world_size = get_dist_info()
    os.makedirs(tmpdir, exist_ok=True)
    dist.barrier()
    with open(os.path.join(tmpdir, 'result_part_{}.pkl'.format(rank)), 'wb') as f:
        pickle.dump(result_part, f)
    dist.barrier()
    if rank != 0:
        return None
    part_list = []
    for i in range(world_size):
        with open(os.path.join(tmpdir, 'result_part_{}.pkl'.format(i)), 'rb') as f:
            part_list.append(pickle.load(f))
    ordered = interleave_parts(part_list, size)
    shutil.rmtree(tmpdir)
    return ordered
