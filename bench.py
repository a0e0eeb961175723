#!/usr/bin/env python
"""Headline benchmark: LiDAR frames/sec of the per-frame detection path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the whole hot path over one batch of --batch synthetic 160k-point frames
(BASELINE.json configs[1]: 0.1 m voxels, grid 1504x1504x40, full VoxelResBackBone8x + BaseBEVBackbone +
CenterHead + decode + rotated NMS; the reference evaluates with BATCH_SIZE_PER_GPU frames per pass the same
way), frames already resident in HBM when the timed region starts.  --math f32 runs every convolution on
the fp32 matrix cores (exact fp32); the default f16x2 carries every fp32 value as an (hi, lo) pair of fp16
and evaluates products as three fp16 MFMAs with fp32 accumulation (csrc/hgemm.h: 22-bit significands, head
maps within 5e-7 of the fp32 path, boxes within the 1e-3 the north star asks for - tests/test_gpu_split.py).  One process per GPU, frames sharded across GPUs (weak scaling); with N>1 the per-frame boxes are
gathered to rank 0 with one RCCL all-gather at the end of the timed region.  value = frames/s over all GPUs.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: fp32 MFMA peak (dense)
PEAK_F16_MFMA_TFLOPS = 2516.6    # same guide: fp16 / bf16 MFMA dense (256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz)
PEAK_HBM_GBS = 8000.0            # HBM3E spec


def usable_cores():
    """Host cores this process may really use: affinity mask and cgroup CPU quota, not os.cpu_count()
    (a container limited to a few CPUs on a many-core host otherwise oversubscribes OpenMP 10-100x)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def pmc_traffic(kernel, math):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC pass (profiles/*_pmc_traffic.json,
    written by tools/gpu_round.sh full from separate FETCH_SIZE / WRITE_SIZE passes over this same command):
    FETCH_SIZE is in KiB and counts 64 B per 128-B request on gfx950 (x2, MI355X_MICROARCH.md section HBM); WRITE_SIZE is
    taken as reported.  None when no profile of this kernel is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json')))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        tag = {'f16x2': '[F16]', 'bf16x2': '[BF16]'}.get(math, '')
        ent = d.get(kernel + tag) or d.get(kernel.replace('>', ' >'))
        if not ent or 'FETCH_SIZE' not in ent:
            return None
        rd = 2.0 * 1024.0 * ent['FETCH_SIZE']['per_call']
        wr = 1024.0 * ent.get('WRITE_SIZE', {'per_call': 0.0})['per_call']
        return {'read_bytes': round(rd), 'write_bytes': round(wr), 'bytes': round(rd + wr), 'source': os.path.basename(files[-1])}
    except Exception:
        return None


def log(*a):
    print('[bench %.1fs]' % (time.perf_counter() - _T0), *a, file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--points', type=int, default=160000)
    ap.add_argument('--batch', type=int, default=16, help='frames per step per GPU (reference eval: BATCH_SIZE_PER_GPU)')
    ap.add_argument('--math', default='f16x2', choices=['f32', 'f16x2', 'bf16x2'],
                    help='conv arithmetic: f32 = fp32 MFMA; f16x2 / bf16x2 = split-precision pairs on the 16-bit matrix cores')
    ap.add_argument('--overlap', action='store_true',
                    help='two-stage streaming pipeline (StreamingDetector) instead of one graph per step; measured slower on MI355X')
    ap.add_argument('--no-calibrate', action='store_true', help='keep worst-case level capacities (limits the batch to ~4 frames)')
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a HIP graph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-seconds', type=float, default=12.0)
    ap.add_argument('--profile-frames', type=int, default=3, help='eager passes timed per launch for the roofline')
    return ap.parse_args()


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', rank=rank, world_size=world)
    if args.gpus != world:
        if rank == 0:
            print('note: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE' % (args.gpus, world), file=sys.stderr)
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    from detzero_amd import ops
    from detzero_amd.centerpoint import FramePipeline, synth_detector
    from detzero_amd.synth import VOXEL_SIZE_01, synth_waymo_frame
    from detzero_amd import frame_parallel as fp

    torch.set_num_threads(min(usable_cores(), 32))
    log('rank', rank, 'of', world, 'usable host cores', usable_cores())
    model, cfg, info = synth_detector(VOXEL_SIZE_01, seed=0)
    model = model.to(dev)
    pipe = FramePipeline(model, info, math=args.math)
    B = max(1, args.batch)
    n_distinct = max(4, B + 1)
    frames = [torch.from_numpy(synth_waymo_frame(1000 * rank + i, args.points)).to(dev) for i in range(n_distinct)]
    pool = torch.stack(frames + frames[:B], dim=0)          # (n_distinct + B, N, C): every window of B frames is contiguous
    static_in = pool[:B].clone()                            # (B, N, C) static input of the captured graph
    if not args.no_calibrate:
        caps = pipe.calibrate(frames)           # row capacities of the deep sparse levels from the sample frames (x1.5)
        log('calibrated level capacities per frame:', caps)
    K, W = args.steps, args.warmup
    post_max = pipe.post_max
    results = torch.zeros((K, B, post_max, 9), dtype=torch.float32, device=dev)
    counts = torch.zeros((K, B), dtype=torch.int32, device=dev)

    def load_inputs(i):
        o = (i * B) % n_distinct
        static_in.copy_(pool[o:o + B], non_blocking=True)   # one device-to-device copy of the B frames of this step

    # warm-up (also primes the caching allocator and builds the kernel-layout weights)
    use_graph = not args.no_graph
    graph = None
    streamer = None
    g_out = g_n = None
    if args.overlap:
        from detzero_amd.centerpoint import StreamingDetector
        streamer = StreamingDetector(pipe, frames[:B], use_graph=use_graph, warmup=max(W, 3))   # (per-frame input slots)
        graph_note = streamer.graph_note
    else:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(max(W, 3)):
                load_inputs(i)
                g_out, g_n = pipe(static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph_note = 'hipGraph replay'
        if use_graph:
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    g_out, g_n = pipe(static_in)
                graph.replay()
                torch.cuda.synchronize()
            except Exception as e:  # capture is an optimisation, not a requirement
                graph = None
                graph_note = 'eager launches (graph capture failed: %s)' % str(e).split('\n')[0][:120]
                torch.cuda.synchronize()
        else:
            graph_note = 'eager launches'

    def batch_of(i):
        return [frames[(i * B + j) % n_distinct] for j in range(B)]

    def step(i):
        """One step = one batch through the whole path.  Streaming mode: stage A of batch i is enqueued together with
        stage B of batch i-1 (whose results land in slot i-1); exactly one A and one B per call."""
        nonlocal g_out, g_n
        if streamer is not None:
            prev = streamer.feed(batch_of(i))
            if prev is not None:
                results[(i - 1) % K].copy_(prev[0], non_blocking=True)
                counts[(i - 1) % K].copy_(prev[1], non_blocking=True)
            return
        load_inputs(i)
        if graph is not None:
            graph.replay()
        else:
            g_out, g_n = pipe(static_in)
        results[i % K].copy_(g_out, non_blocking=True)
        counts[i % K].copy_(g_n, non_blocking=True)

    log('launch mode:', graph_note)
    for i in range(W + 1):            # streaming mode: primes the pipeline (the first feed has no stage B)
        step(i)
    torch.cuda.synchronize()
    log('warm-up done')
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(W + 1, W + 1 + K):
        step(i)
    if world > 1:
        all_b, all_c = fp.gather_frame_boxes(results.view(K * B, post_max, 9), counts.view(K * B))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if pipe.last_overflow is not None and bool(pipe.last_overflow.item()):
        raise SystemExit('bench: a sparse level overflowed its calibrated capacity - rerun with --no-calibrate')
    n_boxes = counts.float().mean().item()
    log('timed region: %d steps x %d frames in %.3f s' % (K, B, dt))

    out = None
    dtype_name = {'f32': 'f32', 'f16x2': 'f32 as f16 pairs (hi+lo, 22-bit significand; 3 f16 MFMA per product, f32 accumulate)',
                  'bf16x2': 'f32 as bf16 pairs (hi+lo, 16-bit significand; 3 bf16 MFMA per product, f32 accumulate)'}[args.math]
    if rank == 0:
        value = world * K * B / dt
        out = {
            'metric': 'LiDAR frames/sec (160k pts, 0.1m voxels)', 'value': round(value, 3), 'unit': 'frames/s',
            'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': round(1000.0 * dt / K, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype_name, 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: %d-pt synthetic Waymo frames, 0.1 m voxels '
                                   '(grid 1504x1504x40), hard voxelize + MeanVFE + VoxelResBackBone8x + BaseBEVBackbone '
                                   '+ CenterHead + decode + rotated NMS, frames resident in HBM' % args.points,
                       'frames_per_step_per_gpu': B, 'ms_per_frame': round(1000.0 * dt / (K * B), 4), 'parallelism': 'frame-parallel x%d' % world,
                       'launch': graph_note, 'math': args.math, 'overlap': 'stage A (voxelize + index pyramid) of batch i+1 under stage B (convs, head, NMS) of batch i' if streamer is not None else 'none', 'weights': 'seeded random init (no checkpoints offline)',
                       'mean_boxes_per_frame': round(n_boxes, 1)},
        }

    # ---- roofline of the dominant kernel: HIP events around every conv launch, on the launch stream
    if rank == 0:
        prof = ops.LaunchProfiler()
        ops.PROFILER = prof
        stage_ms = {}
        try:
            for i in range(args.profile_frames):
                load_inputs(i)
                pipe(static_in)
            agg = prof.summary()
        finally:
            ops.PROFILER = None
        kern = []
        for name, a in agg.items():
            per = a['ms'] / a['launches']
            kern.append({'kernel': name, 'launches_per_step': a['launches'] / args.profile_frames,
                         'avg_us': round(1000.0 * per, 2), 'ms_per_step': round(a['ms'] / args.profile_frames, 4),
                         'tflops': round(a['flops'] / (a['ms'] * 1e-3) / 1e12, 2),
                         'algorithmic_gbs': round(a['bytes'] / (a['ms'] * 1e-3) / 1e9, 1)})
        kern.sort(key=lambda r: -r['ms_per_step'])
        if kern:
            top = kern[0]
            a = agg[top['kernel']]
            achieved = a['flops'] / (a['ms'] * 1e-3) / 1e12
            split = '_h<' in top['kernel']
            # split engine: three 16-bit MFMAs per algorithmic product -> algorithmic peak = f16 MFMA peak / 3
            peak = PEAK_F16_MFMA_TFLOPS / 3.0 if split else PEAK_F32_MFMA_TFLOPS
            out['roofline'] = {'bound': 'mfma', 'kernel': top['kernel'], 'achieved': round(achieved, 2),
                               'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
                               'traffic': pmc_traffic(top['kernel'], args.math),
                               'flop_per_launch': round(a['flops'] / a['launches'], 1),
                               'avg_launch_us': round(1000.0 * a['ms'] / a['launches'], 2),
                               'note': ('split-precision pairs: 3 x v_mfma_f32_32x32x16_f16 per product, peak = 2516.6/3 TF/s '
                                        'algorithmic; ' if split else 'fp32-input MFMA (v_mfma_f32_16x16x4_f32), peak 157.3 TF/s dense; ')
                                       + 'algorithmic FLOP = 2*pixels*taps*Cin*Cout (dense) / 2*pairs*Cin*Cout (sparse); '
                                         'fraction of the fp32-MFMA peak: %.3f' % (achieved / PEAK_F32_MFMA_TFLOPS)}
        out['kernels'] = kern
        log('per-kernel profile done')
        out['conv_ms_per_frame'] = round(sum(r['ms_per_step'] for r in kern) / B, 4)

    # ---- CPU baseline: the oracle (reference-semantics restatement) on this host's cores, bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from tests.util import cpu_state_dict, oracle_detect       # oracle = checker/baseline only
        cores = min(usable_cores(), 32)
        torch.set_num_threads(cores)
        sd = cpu_state_dict(model)
        pts = [f.cpu().numpy() for f in frames]
        from oracle.voxelize import mask_points_by_range
        done, t_cpu = 0, 0.0
        while t_cpu < args.cpu_baseline_seconds and done < 8:
            p = pts[done % n_distinct]
            p = p[mask_points_by_range(p, info.point_cloud_range)]
            t1 = time.perf_counter()
            oracle_detect(sd, p, info)
            t_cpu += time.perf_counter() - t1
            done += 1
        out['cpu_baseline'] = {'value': round(done / t_cpu, 4), 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
                               'sample': '%d frames of the same 160k-pt workload through oracle/ (numpy + CPU torch '
                                         'restatement of the spconv/PyTorch path; spconv itself is not installable), '
                                         '%.1f s' % (done, t_cpu)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
