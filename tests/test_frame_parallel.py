"""CPU (gloo, world_size 2): frame sharding identical to the reference's DistributedSampler and the
padded box gather + rank-0 re-interleave (replaces merge_results_dist's pickle files)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from detzero_amd import frame_parallel as fp


def test_shard_indices_match_reference_sampler():
    # reference: indices += indices[:total-len]; indices[rank:total:world]  (datasets/__init__.py:29-33)
    for n, world in [(10, 2), (7, 4), (200, 8), (3, 8), (1, 2)]:
        total = ((n + world - 1) // world) * world
        parts = [fp.shard_indices(n, r, world) for r in range(world)]
        assert all(len(p) == total // world for p in parts)
        merged = fp.interleave_parts(parts, n)
        assert merged == list(range(n))
        if total - n <= n:
            ref = list(range(n)) + list(range(n))[:total - n]
            assert parts == [ref[r:total:world] for r in range(world)]


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_frames, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)

    class Frames:
        num_frames = n_frames

        def __call__(self, i):
            return torch.full((4, 5), float(i))

    def pipeline(pts):          # fake detector: frame id encoded in the boxes, count = id % 3 + 1
        i = int(pts[0, 0].item())
        b = torch.zeros((6, 9))
        b[:, 0] = i
        b[:, 7] = 0.5
        b[:, 8] = 1 + (i % 3)
        return b, torch.tensor([i % 3 + 1], dtype=torch.int32)
    def batched(frames):      # the same detector over a list of frames -> (B,K,9), (B,)
        outs = [pipeline(p) for p in frames]
        return torch.stack([o[0] for o in outs]), torch.cat([o[1] for o in outs])
    res = fp.run_frame_parallel(pipeline, Frames(), ['Vehicle', 'Pedestrian', 'Cyclist'],
                                metas=[{'frame_id': i} for i in range(n_frames)])
    res_b = fp.run_frame_parallel(batched, Frames(), ['Vehicle', 'Pedestrian', 'Cyclist'],
                                  metas=[{'frame_id': i} for i in range(n_frames)], batch=3)
    # bench.py's multi-GPU structure: sharded steps, gather inside the timed region, MAX over ranks of the elapsed time
    K, B = 3, 2
    results = torch.zeros((K, B, 6, 9)); counts = torch.zeros((K, B), dtype=torch.int32)
    calls = []

    def step(i):
        calls.append(i)
        results[i % K, :, :, 0] = 100 * rank + i
        counts[i % K] = rank + 1
        if rank == 1 and i >= 2:
            import time
            time.sleep(0.05)          # the slow rank decides the time
    dt, all_b, all_c = fp.timed_steps(step, K, 2, results, counts, sync=lambda: None)
    if rank == 0:
        same = all(a['frame_id'] == b['frame_id'] and np.array_equal(a['boxes_lidar'], b['boxes_lidar']) and list(a['name']) == list(b['name'])
                   for a, b in zip(res, res_b))
        timed = (calls == [0, 1, 2, 3, 4], dt >= 0.15, tuple(all_b.shape) == (2, K * B, 6, 9), all_c[1].tolist() == [2] * (K * B),
                 float(all_b[1, 0, 0, 0]) == 103.0, float(all_b[0, 0, 0, 0]) == 3.0)
        q.put(([(r['frame_id'], r['boxes_lidar'].shape[0], float(r['boxes_lidar'][0, 0]), str(r['name'][0])) for r in res], same, timed))
    else:
        assert res_b is None and all_b is None
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_gather_and_order():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    n = 7
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    out, same, timed = q.get(timeout=120)
    assert same, 'batched run differs from the frame-by-frame run'
    assert all(timed), timed
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    names = ['Vehicle', 'Pedestrian', 'Cyclist']
    assert [o[0] for o in out] == list(range(n))
    for i, (fid, cnt, x0, name) in enumerate(out):
        assert cnt == i % 3 + 1 and x0 == float(i) and name == names[i % 3]


def test_boxes_to_annos_format():
    b = torch.zeros((5, 9)); b[0, :7] = torch.arange(7.0); b[0, 7] = 0.9; b[0, 8] = 2
    a = fp.boxes_to_annos(b, 1, ['Vehicle', 'Pedestrian', 'Cyclist'], {'sequence_name': 's', 'frame_id': 3})
    assert a['name'][0] == 'Pedestrian' and a['boxes_lidar'].shape == (1, 7) and a['score'].shape == (1,)
    assert a['sequence_name'] == 's' and a['frame_id'] == 3
    e = fp.boxes_to_annos(b, 0, ['Vehicle'])
    assert e['boxes_lidar'].shape[0] == 0


def _merge_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import detzero_amd.shim as shim
    shim.install()
    from detzero_utils import common_utils
    n = 7                                               # 7 frames over 2 ranks: rank 1's shard ends with the wrap-around frame 0
    part = [{'frame_id': i, 'boxes_lidar': np.full((2, 7), i, np.float32)} for i in fp.shard_indices(n, rank, world)]
    merged = common_utils.merge_results_dist(part, n, tmpdir='/nonexistent/never-created')
    assert common_utils.get_dist_info() == (rank, world)
    q.put((rank, None if merged is None else [m['frame_id'] for m in merged]))
    dist.barrier()
    dist.destroy_process_group()


def test_merge_results_dist_is_a_collective_world2():
    """The shim's merge_results_dist (the name eval_utils.eval_one_epoch calls) over gloo, world size 2: rank 0 gets the records in
    dataset order with the sampler's wrap-around padding cut, rank 1 gets None, and no file is written (the reference's version
    goes through per-rank pickle files, common_utils.py:119-140)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_merge_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == list(range(7)) and got[1] is None
    assert not os.path.exists('/nonexistent')
