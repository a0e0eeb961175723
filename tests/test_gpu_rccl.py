"""GPU (MI355X): the multi-GPU plumbing on the backend it ships with.  ``torch.distributed``'s ``nccl`` backend IS RCCL on ROCm; a
one-GPU box cannot host two ranks (RCCL refuses two ranks on one device), so this runs the 8-GPU job's code path with ONE rank:
process-group init over 127.0.0.1, the all-reduced overflow flag, ``all_gather_into_tensor`` of the padded boxes / counts, the
rank-0 re-interleave, and bench.py's timed region - on device tensors produced by the HIP detector.  The world-size-2 logic is
covered on CPU by tests/test_frame_parallel.py (gloo); reference: detection/detzero_det/datasets/__init__.py:16-36 (sampler),
utils/detzero_utils/common_utils.py:119-140 (merge)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, socket, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
from detzero_amd import frame_parallel as fp
from detzero_amd.centerpoint import FramePipeline
from detzero_amd.synth import VOXEL_SIZE_02
from tests.util import make_model, masked_frame
model, cfg, info = make_model(VOXEL_SIZE_02, seed=0)
pipe = FramePipeline(model.to(dev), info)
frames = [torch.from_numpy(masked_frame(s, 20000)).to(dev) for s in (0, 3, 5)]
names = ['Vehicle', 'Pedestrian', 'Cyclist']
plain = fp.run_frame_parallel(pipe, frames, names, batch=2)                 # no process group: no collective
dist.init_process_group('nccl', rank=0, world_size=1)
assert dist.get_backend() == 'nccl'
got = fp.run_frame_parallel(pipe, frames, names, batch=2)                   # all_reduce of the flag + all_gather_into_tensor (RCCL)
assert len(got) == len(plain) == 3
for a, b in zip(plain, got):
    assert np.array_equal(a['boxes_lidar'], b['boxes_lidar']) and np.array_equal(a['score'], b['score']) and list(a['name']) == list(b['name'])
assert sum(len(a['score']) for a in got) > 0
boxes = torch.arange(2 * 4 * 9, dtype=torch.float32, device=dev).view(2, 4, 9)
counts = torch.tensor([4, 1], dtype=torch.int32, device=dev)
ab, ac = fp.gather_frame_boxes(boxes, counts)
torch.cuda.synchronize()
assert tuple(ab.shape) == (1, 2, 4, 9) and torch.equal(ab[0], boxes) and torch.equal(ac[0], counts)
K, B = 2, 2
res = torch.zeros((K, B, 4, 9), device=dev); cnt = torch.zeros((K, B), dtype=torch.int32, device=dev)
def step(i):
    res[i %% K] = float(i); cnt[i %% K] = i
info_t = {}
dt, _, _ = fp.timed_steps(step, K, 1, res, cnt, sync=torch.cuda.synchronize, info=info_t)
assert dt > 0 and info_t['ranks_seen'] == 1
dist.barrier()
dist.destroy_process_group()
print('RCCL_WORLD1_OK')
'''


def test_frame_parallel_on_rccl_world_size_one(device):
    out = subprocess.run([sys.executable, '-c', WORKER % {'root': ROOT}], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and 'RCCL_WORLD1_OK' in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])
