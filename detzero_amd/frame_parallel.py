"""Frame-parallel multi-GPU inference: one process per GPU, frames sharded exactly like the
reference's no-shuffle DistributedSampler, per-frame boxes gathered to rank 0 with ONE padded
collective (RCCL over xGMI; ``gloo`` on CPU for tests).

Reference:
  * sampler   - /root/reference/detection/detzero_det/datasets/__init__.py:16-36
  * result merge - /root/reference/utils/detzero_utils/common_utils.py:119-140 (pickle files +
    two barriers; here the payload travels in a collective instead)
  * result record format - detection/detzero_det/datasets/dataset.py:305-354, consumed by
    tracking/detzero_track/datasets/waymo_dataset.py:51-63

Payload per frame: (K=500, 9) fp32 ``[x,y,z,dx,dy,dz,heading,score,label]`` + one int32 count =
18 KB, i.e. 3.6 MB for a 200-frame sequence: latency-bound, far below one xGMI link (~153 GB/s),
so the gather is issued once per chunk of frames, not per frame.
"""
import numpy as np
import torch
import torch.distributed as dist

from .lib import DetZeroHipError


def shard_indices(num_frames, rank, world_size):
    """Indices rank ``rank`` processes: pad by wrap-around to a multiple of the world size, then
    stride (datasets/__init__.py:23-34 with shuffle=False)."""
    indices = list(range(num_frames))
    total = ((num_frames + world_size - 1) // world_size) * world_size
    while len(indices) < total:                       # wrap-around padding, also when total > 2*num_frames
        indices += indices[:total - len(indices)]
    return indices[rank:total:world_size]


def interleave_parts(parts, size):
    """rank-0 re-ordering of merge_results_dist (common_utils.py:135-138): zip(*parts), truncate."""
    ordered = []
    for res in zip(*parts):
        ordered.extend(list(res))
    return ordered[:size]


def gather_frame_boxes(boxes, counts, group=None, dst=0):
    """boxes (F,K,9) float32, counts (F,) int32 on every rank (same F) -> on ``dst``:
    (world,F,K,9), (world,F); ``None`` elsewhere.  One all_gather each for boxes and counts
    Why all-gather and not gather: the payload is 18 KB per frame (latency-bound either way); RCCL implements
    all_gather_into_tensor as ONE ring collective over xGMI, whereas gather() is a group of point-to-point send/recv
    pairs into rank 0 (W-1 transfers serialised on rank 0's links) that also needs per-rank output lists; the copies the
    other ranks receive are simply dropped."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    f, k, c = boxes.shape
    out_b = torch.empty((world, f, k, c), dtype=boxes.dtype, device=boxes.device)
    out_c = torch.empty((world, f), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(out_b.view(world * f, k, c), boxes.contiguous(), group=group)
    dist.all_gather_into_tensor(out_c.view(world * f), counts.contiguous(), group=group)
    if rank != dst:
        return None, None
    return out_b, out_c


def boxes_to_annos(boxes9, count, class_names, frame_meta=None):
    """One frame's padded boxes -> the reference's prediction dict (dataset.py:325-352)."""
    n = int(count)
    b = np.asarray(boxes9[:n].detach().cpu().numpy() if torch.is_tensor(boxes9) else boxes9[:n])
    anno = {
        'name': np.array(class_names)[b[:, 8].astype(np.int64) - 1] if n else np.zeros(0),
        'score': b[:, 7].copy() if n else np.zeros(0),
        'boxes_lidar': b[:, :7].copy() if n else np.zeros([0, 9]),
    }
    if frame_meta:
        anno.update(frame_meta)
    return anno


def run_frame_parallel(pipeline, frames, class_names, group=None, metas=None, batch=1):
    """Run ``pipeline`` over this rank's shard of ``frames`` (a list of device tensors or a callable index -> tensor with a
    ``num_frames`` attribute), gather, and return on rank 0 the list-of-dicts result in dataset order (what result.pkl
    holds); other ranks return None.

    batch = frames per pipeline call (the reference's BATCH_SIZE_PER_GPU).  batch == 1: ``pipeline(points) -> ((K,9), (1,))``;
    batch > 1: ``pipeline([points, ...]) -> ((B,K,9), (B,))`` - the shard is walked in consecutive groups of ``batch`` frames
    (sampler order is kept: only the grouping changes), the last group may be shorter."""
    grouped = dist.is_initialized()     # an initialised process group always goes through the collectives, also with ONE rank (the
                                        # single-GPU RCCL test runs exactly the code path of the 8-GPU job)
    world = dist.get_world_size(group) if grouped else 1
    rank = dist.get_rank(group) if grouped else 0
    n = len(frames) if not callable(frames) else frames.num_frames
    mine = shard_indices(n, rank, world)
    get = frames if callable(frames) else frames.__getitem__
    outs, cnts = [], []
    if batch <= 1:
        for i in mine:
            b, c = pipeline(get(i))
            outs.append(b[None])
            cnts.append(c.reshape(1))
    else:
        for s0 in range(0, len(mine), batch):
            b, c = pipeline([get(i) for i in mine[s0:s0 + batch]])
            outs.append(b)
            cnts.append(c.reshape(-1))
    boxes = torch.cat(outs, dim=0)
    counts = torch.cat(cnts, dim=0).to(torch.int32)
    if hasattr(pipeline, 'overflow_seen'):
        # calibrated level capacities: a dropped site must not go unnoticed (one sync per chunk).  With several ranks the flag is
        # all-reduced BEFORE anyone raises: a rank that raised on its own would leave the others waiting in the box gather below
        # until the collective times out
        over = bool(pipeline.overflow_seen())
        if grouped:
            flag = torch.tensor([1 if over else 0], dtype=torch.int32, device=boxes.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            over = bool(flag.item())
        if over:
            raise DetZeroHipError('run_frame_parallel: a sparse level overflowed its calibrated row capacity on some rank (per-frame '
                                  'capacities %s): re-run calibrate() on denser samples / with a larger margin, or drop the calibration'
                                  % (getattr(pipeline, 'level_caps', None),))
    if not grouped:
        all_b, all_c = boxes[None], counts[None]
    else:
        all_b, all_c = gather_frame_boxes(boxes, counts, group)
        if rank != 0:
            return None
    all_b, all_c = all_b.cpu(), all_c.cpu()
    parts = [[(all_b[r, j], all_c[r, j]) for j in range(all_b.shape[1])] for r in range(world)]
    ordered = interleave_parts(parts, n)
    return [boxes_to_annos(b, c, class_names, metas[i] if metas else None) for i, (b, c) in enumerate(ordered)]


def timed_steps(step, steps, warmup, results, counts, sync, group=None, info=None):
    """The timed region of bench.py, shared with the CPU (gloo) test of the multi-GPU plumbing: ``warmup`` untimed calls of
    ``step(i)``, then exactly ``steps`` timed calls bracketed by barrier + ``sync()`` on both sides; with more than one rank
    the per-frame boxes in ``results`` (steps, B, K, 9) / ``counts`` (steps, B) are gathered inside the timed region
    (the tracker needs them on rank 0) and the elapsed time is the MAX over ranks.
    Returns (seconds, gathered boxes (world, steps*B, K, 9) or None, gathered counts or None) - gathered tensors on rank 0 only.
    info (optional dict) receives ``ranks_seen`` (an all-reduced count of the ranks that ran the region) and ``gather_ms``
    (this rank's time from the end of its last step to the end of the box gather, inside the timed region)."""
    import time
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    for i in range(warmup):
        step(i)
    sync()
    if world > 1:
        dist.barrier(group)
    sync()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        step(i)
    all_b = all_c = None
    t_g = 0.0
    if world > 1:
        sync()
        tg0 = time.perf_counter()
        k, b = results.shape[0], results.shape[1]
        all_b, all_c = gather_frame_boxes(results.view(k * b, results.shape[2], results.shape[3]), counts.view(k * b), group)
        sync()
        t_g = time.perf_counter() - tg0
    sync()
    if world > 1:
        dist.barrier(group)
    sync()
    dt = time.perf_counter() - t0
    seen = 1
    if world > 1:
        t = torch.tensor([dt, t_g], dtype=torch.float64, device=results.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        dt, t_g = float(t[0].item()), float(t[1].item())
        one = torch.ones((1,), dtype=torch.int32, device=results.device)
        dist.all_reduce(one, op=dist.ReduceOp.SUM, group=group)
        seen = int(one.item())
    if info is not None:
        info['ranks_seen'] = seen
        info['gather_ms'] = round(1000.0 * t_g, 3)
    return dt, all_b, all_c
