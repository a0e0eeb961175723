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
maps within 5e-7 of the fp32 path; boxes within the north star's 1e-3 of the CPU oracle ON THIS WORKLOAD -
tests/test_gpu_full_parity.py).  One process per GPU, frames sharded across GPUs (weak scaling); with N>1 the
per-frame boxes are gathered to rank 0 with one RCCL all-gather inside the timed region.
value = frames/s over all GPUs.  Prints ONE JSON line on rank 0.

Besides the headline the default single-GPU run measures, into the same JSON line (each a few seconds; --no-aux skips):
  fp32       the same workload with every convolution in exact fp32 (--fp32-batch frames per step), with its own roofline
  ref_batch  the headline arithmetic at the reference's BATCH_SIZE_PER_GPU = 8 (centerpoint_1sweep.yaml:88)
  batch16    ... at 16 frames per pass, the default of rounds 1-3 (the default is 32 since round 4: DESIGN.md section 4, batch sweep)
  ragged     frames of 150k-180k points: padded to the slot capacity with out-of-range rows (the stacked route), and as a
             ragged list (per-frame voxelizers on parallel streams)
  f16        opt-in fast mode: one fp16 MFMA per product on the same fp16-pair tensors (not fp32-class; its own tolerance), with roofline
  multisweep BASELINE configs[4] shape: two merged sweeps per frame (320k points, 6 features), DynamicMeanVFE, 3-sweep model
  with_h2d   frames start in pinned host memory; the H2D copy of step i+1 runs on a copy stream under step i
  stages     voxelize / index pyramid / sparse backbone / dense / post-processing: time per step (each stage replayed as its
             own hipGraph, serially), algorithmic bytes and HBM GB/s as a fraction of the 8 TB/s peak
  gather     the gather sparse engine (default of rounds 1-3) on the headline workload: A/B against the x-run engine on this box
  tiles      (--tiles-leg) the opt-in tile-resident sparse engine (csrc/sparse_conv_t.hip, rows in brick order)
  refine     BASELINE configs[3], the secondary kernel set: GRM objects/s and PRM tracks/s (fp32 and f16x2 stacks) and the
             attention core (k_mha_block) with its roofline against the fp32-MFMA peak
  pdv        the two-stage detector (PDVHead second stage) on merged 2-sweep frames: ms per stage, RoIs/s, frames/s at 1 / 8 frames per pass
             through the plugin modules and at 8 / 16 through FramePipeline.two_stage
The exact-fp32 leg is also promoted to the top-level keys value_fp32 / roofline_fp32 (the precision-equivalent number next to
`value`, whose arithmetic carries 22 significant bits).

N > 1: `python bench.py --gpus N` starts the N ranks itself (re-executes under torch.distributed.run on 127.0.0.1) when it is not
already running under a launcher, and fails if the node has fewer than N GPUs; the line carries ranks_seen (all-reduced) and the
time of the box gather.
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
REF_BATCH = 8                    # OPTIMIZATION.BATCH_SIZE_PER_GPU of centerpoint_1sweep.yaml:88


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
        tag = {'f16x2': '[F16]', 'bf16x2': '[BF16]', 'f16': '[F16H]'}.get(math, '')
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
    ap.add_argument('--steps', type=int, default=160, help='timed steps (default: ~9 s of GPU time at 64 frames per step)')
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--points', type=int, default=160000)
    ap.add_argument('--batch', type=int, default=64, help='frames per step per GPU (reference eval: BATCH_SIZE_PER_GPU = 8, leg ref_batch; 16 was the default of '
                    'rounds 1-3, leg batch16; 32 of rounds 4-6, leg batch32; a step of 64 runs as two concurrent sub-passes of 32 frames - the size '
                    'one pass is best at, DESIGN.md section 4: 32 / 48 / 64 -> 1088 / 1107 / 1113 frames/s on one box; DetZero labels sequences '
                    'offline: throughput, not latency, is the metric)')
    ap.add_argument('--math', default='f16x2', choices=['f32', 'f16x2', 'bf16x2', 'f16'],
                    help='conv arithmetic: f32 = fp32 MFMA; f16x2 / bf16x2 = split-precision pairs on the 16-bit matrix cores; '
                         'f16 = one fp16 MFMA per product on the same tensors (fast mode, not fp32-class)')
    ap.add_argument('--overlap', action='store_true',
                    help='two-stage streaming pipeline (StreamingDetector: stage A of batch i+1 under stage B of batch i, results one step later) instead of '
                         'one graph per step; +0.8 .. +1.4 %% on MI355X (r03e build, A/B on one box), not the default')
    ap.add_argument('--no-calibrate', action='store_true', help='keep worst-case level capacities (limits the batch to ~4 frames)')
    ap.add_argument('--no-graph', action='store_true', help='launch eagerly instead of replaying a HIP graph')
    ap.add_argument('--weights', default='preserve', choices=['preserve', 'default'],
                    help="synthetic weight set: 'preserve' (round 6: boxes depend on the frame) or 'default' (rounds 1-5: the default initialisers - "
                         'frame-independent border boxes, near-constant activations; for continuity with the earlier lines only)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-seconds', type=float, default=12.0)
    ap.add_argument('--dump-launches', default=None, help='file: every conv launch of one eager pass in launch order (us, GFLOP, MB)')
    ap.add_argument('--profile-frames', type=int, default=3, help='eager passes timed per launch for the roofline')
    ap.add_argument('--no-aux', action='store_true', help='skip the auxiliary legs (fp32, ref_batch, ragged, f16, multisweep, with_h2d, stages)')
    ap.add_argument('--aux-seconds', type=float, default=2.5, help='timed GPU seconds per auxiliary leg')
    ap.add_argument('--fp32-batch', type=int, default=0, help='frames per step of the exact-fp32 leg (0 = the headline batch)')
    ap.add_argument('--tiles-leg', action='store_true', help='also time the opt-in tile-resident engine of round 3 (slower)')
    ap.add_argument('--sparse-engine', default='xrun', choices=['gather', 'xrun', 'tiles'], help='sparse-backbone engine of the headline run')
    ap.add_argument('--sweeps', type=int, default=1, choices=[1, 2],
                    help='2: run the multisweep shape (BASELINE configs[4]) as the main workload - for profiling that leg on its own; the '
                         'metric of the printed line is then NOT the headline one (config.workload says so)')
    ap.add_argument('--no-refine', action='store_true', help='skip the refiner leg (BASELINE configs[3])')
    ap.add_argument('--no-pdv', action='store_true', help='skip the two-stage (PDV) leg')
    ap.add_argument('--stub', action='store_true',
                    help='launch-structure test without a GPU: gloo ranks on CPU, a no-op step; exercises the self-spawn, the barrier / '
                         'gather / max-over-ranks timed region and the JSON line (tests/test_bench_contract.py), measures nothing')
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1)
    by re-executing under torch.distributed.run - the same command line the driver uses.  Fails loudly when the node has fewer GPUs."""
    import socket
    if not args.stub:
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit('bench.py: --gpus %d but this node exposes %d GPU(s)' % (args.gpus, have))
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log('starting %d ranks: %s' % (args.gpus, ' '.join(cmd[1:9])))
    os.execvp(cmd[0], cmd)


class Case:
    """One configured pipeline over one batch shape: seeded detector, synthetic frames resident in HBM, calibrated level
    capacities, and the step captured as a hipGraph over a static input.

    lengths = None: every frame has `points` points (stacked (B,N,C) input, dz_voxelize_to_level route).
    lengths = (lo, hi), mode 'padded': frames of lo..hi points padded with out-of-range rows to slots of hi rows (same route).
    lengths = (lo, hi), mode 'list': slot j holds a frame of its own length L_j in lo..hi (ragged list route)."""

    def __init__(self, args, dev, rank, math, batch, lengths=None, mode='stacked', seed_base=0, sweeps=1, engine=None, select=False):
        engine = engine or args.sparse_engine
        self.math_selected, self.activation_peaks = math, None
        from detzero_amd.centerpoint import FramePipeline, set_sparse_engine, synth_detector
        from detzero_amd.synth import VOXEL_SIZE_01, merge_two_sweeps, synth_waymo_frame
        self.args, self.dev, self.math, self.B, self.mode = args, dev, math, max(1, batch), mode
        # sweeps = 2: BASELINE configs[4] shape - two sweeps merged into one 6-feature frame (time-offset column), the
        # centerpoint_3sweeps model with DynamicMeanVFE
        self.model, self.cfg, self.info = synth_detector(VOXEL_SIZE_01, seed=0, sweeps=3 if sweeps > 1 else 1, gain=getattr(args, 'weights', 'preserve'))
        self.model = self.model.to(dev)
        set_sparse_engine(self.model, engine)
        self.engine = engine
        self.pipe = FramePipeline(self.model, self.info, dynamic=sweeps > 1, math=math)
        B = self.B
        self.n_distinct = n_distinct = max(4, B + 1)
        rng = np.random.default_rng(77 + rank)
        if lengths is None:
            ns = [args.points] * n_distinct
        elif mode == 'list':
            # a captured graph replays fixed shapes: slot j always holds a frame of ITS length L_j; two groups of B frames alternate
            slot_len = [int(v) for v in rng.integers(lengths[0], lengths[1] + 1, size=B)]
            ns = slot_len + slot_len
            self.n_distinct = n_distinct = 2 * B
        else:
            ns = [int(v) for v in rng.integers(lengths[0], lengths[1] + 1, size=n_distinct)]
        self.host_frames = [synth_waymo_frame(seed_base + 1000 * rank + i, n) for i, n in enumerate(ns)]
        if sweeps > 1:
            self.host_frames = [merge_two_sweeps(f, synth_waymo_frame(seed_base + 1000 * rank + 300 + i, f.shape[0]))
                                for i, f in enumerate(self.host_frames)]
        self.mean_points = float(np.mean([f.shape[0] for f in self.host_frames]))
        if mode == 'list':
            self.frames = [torch.from_numpy(f).to(dev) for f in self.host_frames]
            self.static_in = [self.frames[j].clone() for j in range(B)]
            sample = self.frames[:n_distinct]
        else:
            cap = (args.points if lengths is None else lengths[1]) * (2 if sweeps > 1 else 1)
            padded = np.zeros((n_distinct, cap, self.host_frames[0].shape[1]), np.float32)
            padded[:, :, 0] = 1e6                                        # rows outside POINT_CLOUD_RANGE: dropped by the xy mask
            for i, f in enumerate(self.host_frames):
                padded[i, :f.shape[0]] = f
            self.host_pool = np.concatenate([padded, padded[:B]], axis=0)     # every window of B frames is contiguous
            self.pool = torch.from_numpy(self.host_pool).to(dev)
            self.static_in = self.pool[:B].clone()                       # (B, N, C) static input of the captured graph
            sample = [self.pool[i] for i in range(n_distinct)]
        self.caps = None
        if not args.no_calibrate:
            # row capacities of the deep sparse levels (x1.5 of the largest count seen) from frames DISJOINT from the timed ones
            # (other seeds, same generator and lengths): the overflow flag checked after the timed region is then a real guard
            cal = []
            for i, f in enumerate(self.host_frames[:4]):
                g = synth_waymo_frame(seed_base + 1000 * rank + 7000 + i, f.shape[0] // (2 if sweeps > 1 else 1))
                if sweeps > 1:
                    g = merge_two_sweeps(g, synth_waymo_frame(seed_base + 1000 * rank + 7300 + i, g.shape[0]))
                cal.append(torch.from_numpy(g).to(dev))
            self.caps = self.pipe.calibrate(cal)
            if math in ('f16x2', 'bf16x2') and select:
                # the arithmetic a deployment would pick for this checkpoint (centerpoint.select_math: a calibration pass on the exact-fp32
                # engine, per-stage activation peaks) - reported as config.math_selected; the line is measured in THAT mode
                from detzero_amd.centerpoint import select_math
                self.math_selected, peaks = select_math(self.model, self.info, cal[:2], prefer=math, dynamic=sweeps > 1)
                self.activation_peaks = {k: float('%.4g' % v) for k, v in peaks.items()}
                self.math = self.math_selected            # (select_math has set the model to it)
            del cal
        # the instrumented passes (kernel_profile, stage_profile) run ONE pass, every launch alone on the chip: of the whole batch while it
        # fits one pass (<= 32 frames: the voxel keys of a pass are 32 bits), else of one sub-pass's share of it
        self.PB = B if (B <= 32 or not self.pipe.splits(B)) else B // self.pipe.ways
        self.graph = None
        self.g_out = self.g_n = None
        self.graph_note = 'eager launches'
        self.results = self.counts = None

    def load_inputs(self, i):
        B = self.B
        if self.mode == 'list':
            for j in range(B):
                self.static_in[j].copy_(self.frames[(i % 2) * B + j], non_blocking=True)
        else:
            o = (i * B) % self.n_distinct
            self.static_in.copy_(self.pool[o:o + B], non_blocking=True)   # one device-to-device copy of the B frames of this step

    def prepare(self, steps, warmup, use_graph=True):
        K, B = steps, self.B
        self.results = torch.zeros((K, B, self.pipe.post_max, 9), dtype=torch.float32, device=self.dev)
        self.counts = torch.zeros((K, B), dtype=torch.int32, device=self.dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):        # warm-up (also primes the caching allocator and builds the kernel-layout weights)
            for i in range(max(warmup, 3)):
                self.load_inputs(i)
                self.g_out, self.g_n = self.pipe(self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if use_graph:
            try:
                # (FramePipeline.capture: one graph; a batch run as concurrent sub-passes = parallel branches of it)
                self.graph = self.pipe.capture(self.static_in)
                self.g_out, self.g_n = self.graph.boxes, self.graph.counts
                self.graph.replay()
                torch.cuda.synchronize()
                self.graph_note = 'hipGraph replay' + (' (%d parallel branches: concurrent sub-passes of %d frames)' % (self.graph.branches, B // self.graph.branches)
                                                       if self.graph.branches > 1 else '')
            except Exception as e:  # capture is an optimisation, not a requirement
                self.graph = None
                self.graph_note = 'eager launches (graph capture failed: %s)' % str(e).split('\n')[0][:120]
                torch.cuda.synchronize()

    def run(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.g_out, self.g_n = self.pipe(self.static_in)

    def step(self, i):
        """One step = one batch through the whole path: input copy, voxelize ... NMS, results into slot i % K."""
        K = self.results.shape[0]
        self.load_inputs(i)
        self.run()
        self.results[i % K].copy_(self.g_out, non_blocking=True)
        self.counts[i % K].copy_(self.g_n, non_blocking=True)

    def check_overflow(self):
        if self.pipe.overflow_seen():         # sticky over every pass (graph replay included) since the last check
            raise SystemExit('bench: a sparse level overflowed its calibrated capacity - rerun with --no-calibrate')

    def time_steps(self, steps, warmup, step=None):
        from detzero_amd import frame_parallel as fp
        dt, _, _ = fp.timed_steps(step or self.step, steps, warmup, self.results, self.counts, sync=torch.cuda.synchronize)
        self.check_overflow()
        return dt

    def aux_leg(self, seconds, step=None):
        """Time ~`seconds` of steps (count chosen from a short probe).  Returns (frames/s, ms per step, steps)."""
        if self.g_out is None:
            self.prepare(8, 3, use_graph=not self.args.no_graph)
        probe = self.time_steps(4, 2, step)
        k = int(min(max(seconds / max(probe / 4, 1e-4), 8), 2000))
        self.results = torch.zeros((k,) + tuple(self.results.shape[1:]), dtype=torch.float32, device=self.dev)
        self.counts = torch.zeros((k, self.B), dtype=torch.int32, device=self.dev)
        dt = self.time_steps(k, 2, step)
        return k * self.B / dt, 1000.0 * dt / k, k

    def kernel_profile(self, passes):
        """HIP events around every conv launch, on the launch stream, over `passes` eager passes."""
        from detzero_amd import ops
        prof = ops.LaunchProfiler()
        ops.PROFILER = prof
        ways, self.pipe.ways = self.pipe.ways, 1       # one pass of B frames: every launch timed ALONE (the timed region runs the batch as
        try:                                           # `ways` concurrent sub-passes, whose launches share the chip)
            for i in range(passes):
                self.load_inputs(i)
                self.pipe(self.static_in[:self.PB])
            agg = prof.summary()
            if getattr(self.args, 'dump_launches', None):
                # every launch of the LAST eager pass, in launch order (the per-layer view the aggregated `kernels` list hides)
                per_pass = len(prof.records) // max(passes, 1)
                with open(self.args.dump_launches, 'w') as f:
                    f.write('# launch order of one eager pass (%d frames): kernel, us, algorithmic GFLOP, algorithmic MB, TF/s, GB/s\n' % self.PB)
                    for name, flops, nbytes, e0, e1 in prof.records[-per_pass:]:
                        ms = e0.elapsed_time(e1)
                        f.write('%-34s %9.1f us %9.2f GF %9.1f MB %8.1f TF/s %8.1f GB/s\n' % (name, 1000.0 * ms, flops / 1e9, nbytes / 1e6,
                                                                                           flops / (ms * 1e-3) / 1e12, nbytes / (ms * 1e-3) / 1e9))
        finally:
            ops.PROFILER = None
            self.pipe.ways = ways
        kern = []
        for name, a in agg.items():
            per = a['ms'] / a['launches']
            kern.append({'kernel': name, 'launches_per_step': a['launches'] / passes,
                         'avg_us': round(1000.0 * per, 2), 'ms_per_step': round(a['ms'] / passes, 4),
                         'tflops': round(a['flops'] / (a['ms'] * 1e-3) / 1e12, 2),
                         'algorithmic_gbs': round(a['bytes'] / (a['ms'] * 1e-3) / 1e9, 1)})
        kern.sort(key=lambda r: -r['ms_per_step'])
        roof = None
        if kern:
            top = kern[0]
            a = agg[top['kernel']]
            achieved = a['flops'] / (a['ms'] * 1e-3) / 1e12
            split = '_h<' in top['kernel']
            # split engine: three 16-bit MFMAs per algorithmic product -> algorithmic peak = f16 MFMA peak / 3
            terms = 1.0 if self.math == 'f16' else 3.0
            peak = PEAK_F16_MFMA_TFLOPS / terms if split else PEAK_F32_MFMA_TFLOPS
            roof = {'bound': 'mfma', 'kernel': top['kernel'], 'achieved': round(achieved, 2),
                    'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
                    'flop_per_launch': round(a['flops'] / a['launches'], 1),
                    'avg_launch_us': round(1000.0 * a['ms'] / a['launches'], 2),
                    'note': (('one v_mfma_f32_32x32x16_f16 per product (f16 mode), peak = 2516.6 TF/s; ' if self.math == 'f16' else
                              'split-precision pairs: 3 x v_mfma_f32_32x32x16_f16 per product, peak = 2516.6/3 TF/s algorithmic; ') if split else 'fp32-input MFMA (v_mfma_f32_16x16x4_f32), peak 157.3 TF/s dense; ')
                            + 'algorithmic FLOP = 2*pixels*taps*Cin*Cout (dense) / 2*pairs*Cin*Cout (sparse); '
                              'fraction of the fp32-MFMA peak: %.3f' % (achieved / PEAK_F32_MFMA_TFLOPS)}
        if roof is not None:
            # HBM bytes of the same launch from the committed PMC passes: `traffic` is the scalar the contract names, the parts and
            # the file they come from next to it (flat: nested objects do not survive every consumer of this line)
            tr = pmc_traffic(roof['kernel'], self.math)
            roof['traffic'] = tr['bytes'] if tr else None
            if tr:
                roof.update({'traffic_read_bytes': tr['read_bytes'], 'traffic_write_bytes': tr['write_bytes'], 'traffic_source': tr['source']})
        return kern, roof, agg

    def stage_profile(self, replays=10):
        """Each stage of the step captured as its own hipGraph (index pyramid on the main stream, i.e. serial) and replayed in
        order with HIP events in between.  Returns a list of {stage, ms_per_step, algorithmic_bytes, hbm_gbs, frac_of_hbm_peak}."""
        from detzero_amd.centerpoint import _StackedFrames
        pipe, B = self.pipe, self.PB                      # (one pass: see PB)
        frames = _StackedFrames(self.static_in[:B]) if torch.is_tensor(self.static_in) else self.static_in[:B]
        self.load_inputs(0)
        torch.cuda.synchronize()
        graphs = []
        state = {}

        def capture(fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            graphs.append(g)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            capture(lambda: state.__setitem__('vox', pipe.voxelize_stage(frames)))
            capture(lambda: state.__setitem__('pyr', pipe.pyramid_stage(state['vox'], B, overlap=False)))
            capture(lambda: state.__setitem__('res', pipe.backbone_stage(state['pyr'])))
            capture(lambda: state.__setitem__('head', pipe.dense_stage(state['res'], B)))
            capture(lambda: state.__setitem__('out', pipe.post_stage(*state['head'])))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        names = ['voxelize', 'index', 'sparse_backbone', 'dense', 'post']
        ms = [0.0] * len(graphs)
        for _ in range(replays + 1):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(graphs) + 1)]
            ev[0].record()
            for j, g in enumerate(graphs):
                g.replay()
                ev[j + 1].record()
            torch.cuda.synchronize()
            if _ > 0:
                for j in range(len(graphs)):
                    ms[j] += ev[j].elapsed_time(ev[j + 1]) / replays
        # algorithmic bytes from the live counts: voxelize = points read + voxel rows and coordinates written;
        # index = per rulebook: output coordinates read + 8 B per (in, out) pair written; sparse backbone = sum over the 21 convs
        # of in rows + out rows (+ residual) + weights + 8 B per pair
        pyr = state['pyr']
        steps = pyr['steps']                    # level li: (table of the strided conv INTO it, its submanifold table, SparseLevel, event)
        n_pts = sum(int(f.shape[0]) for f in frames)
        c_in = int(frames[0].shape[1])
        level_m = [st[2].num_active() for st in steps]
        vox_bytes = 4.0 * n_pts * c_in + level_m[0] * (4.0 * c_in + 16.0)
        ch = list(self.model.backbone3d.channels) + [self.model.backbone3d.channels[-1]]     # channels of levels 0..4

        def pairs_of(nbr, m):
            from detzero_amd import ops
            return ops.table_pairs(nbr, m)

        def conv(n_in, cin, m, cout, kvol, pairs, residual):
            return 4.0 * (n_in * cin + m * cout * (2 if residual else 1) + kvol * cin * cout) + 8.0 * pairs
        idx_bytes = conv_bytes = 0.0
        for li, (nbr_d, nbr_s, lvl, _) in enumerate(steps):
            m = level_m[li]
            if nbr_d is not None:               # spconv2/3/4 and conv_out
                pd = pairs_of(nbr_d, m)
                idx_bytes += 16.0 * m + 8.0 * pd
                conv_bytes += conv(level_m[li - 1], ch[li - 1], m, ch[li], nbr_d.shape[0], pd, False)
            if nbr_s is not None:               # conv_input (level 0 only) + two residual blocks
                ps = pairs_of(nbr_s, m)
                idx_bytes += 16.0 * m + 8.0 * ps
                if li == 0:
                    conv_bytes += conv(m, c_in, m, ch[0], 27, ps, False)
                for blk in range(2):
                    conv_bytes += conv(m, ch[li], m, ch[li], 27, ps, False) + conv(m, ch[li], m, ch[li], 27, ps, True)
        algo = {'voxelize': vox_bytes, 'index': idx_bytes, 'sparse_backbone': conv_bytes}
        out = []
        for nme, t in zip(names, ms):
            rec = {'stage': nme, 'frames': B, 'ms_per_step': round(t, 4), 'ms_per_frame': round(t / B, 4)}        # (ms_per_step: per pass of `frames` frames)
            if nme in algo:
                gbs = algo[nme] / (t * 1e-3) / 1e9
                rec.update({'algorithmic_bytes': round(algo[nme]), 'hbm_gbs': round(gbs, 1), 'frac_of_hbm_peak': round(gbs / PEAK_HBM_GBS, 4)})
            out.append(rec)
        return out


def stub_main(args, world, rank):
    """--stub: the launch structure only (gloo ranks on CPU, a no-op step, the shared timed region, the JSON line)."""
    from detzero_amd import frame_parallel as fp
    B, K = 2, args.steps
    results = torch.zeros((K, B, 4, 9), dtype=torch.float32)
    counts = torch.zeros((K, B), dtype=torch.int32)

    def step(i):
        counts[i % K] = rank + 1
    info = {}
    dt, all_b, all_c = fp.timed_steps(step, K, args.warmup, results, counts, sync=lambda: None, info=info)
    if rank == 0:
        print(json.dumps({'metric': 'LiDAR frames/sec (160k pts, 0.1m voxels)', 'value': round(world * K * B / max(dt, 1e-9), 3), 'unit': 'frames/s',
                          'n_gpus': world, 'steps': K, 'warmup': args.warmup, 'ms_per_step': round(1000.0 * dt / K, 4), 'higher_is_better': True,
                          'scaling': 'weak', 'vs_baseline': None, 'dtype': 'none', 'data': 'stub (launch-structure test: no GPU work, measures nothing)',
                          'ranks_seen': info['ranks_seen'], 'gather_ms': info['gather_ms'],
                          'config': {'workload': 'stub', 'frames_per_step_per_gpu': B, 'parallelism': 'frame-parallel x%d' % world,
                                     'gathered_counts': None if all_c is None else [int(v) for v in all_c[:, 0].tolist()]}}), flush=True)


def main():
    args = parse()
    launched = 'WORLD_SIZE' in os.environ
    if args.gpus > 1 and not launched:
        spawn_ranks(args)                        # does not return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('gloo' if args.stub else 'nccl', rank=rank, world_size=world)
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    if args.stub:
        stub_main(args, world, rank)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit('bench.py: rank %d has no GPU (local rank %d, %d visible)' % (rank, local_rank, torch.cuda.device_count()))
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    from detzero_amd import frame_parallel as fp

    torch.set_num_threads(min(usable_cores(), 32))
    log('rank', rank, 'of', world, 'usable host cores', usable_cores())
    case = Case(args, dev, rank, args.math, args.batch, engine=args.sparse_engine, sweeps=args.sweeps, select=True)
    if case.caps is not None:
        log('calibrated level capacities per frame:', case.caps)
    B = case.B
    K, W = args.steps, args.warmup
    streamer = None
    if args.overlap:
        from detzero_amd.centerpoint import StreamingDetector
        case.results = torch.zeros((K, B, case.pipe.post_max, 9), dtype=torch.float32, device=dev)
        case.counts = torch.zeros((K, B), dtype=torch.int32, device=dev)
        frames = [case.pool[i] for i in range(case.n_distinct)]
        streamer = StreamingDetector(case.pipe, frames[:B], use_graph=not args.no_graph, warmup=max(W, 3))   # (per-frame input slots)
        graph_note = streamer.graph_note

        def step(i):
            """Streaming mode: stage A of batch i is enqueued together with stage B of batch i-1 (whose results land in slot
            i-1); exactly one A and one B per call."""
            prev = streamer.feed([frames[(i * B + j) % case.n_distinct] for j in range(B)])
            if prev is not None:
                case.results[(i - 1) % K].copy_(prev[0], non_blocking=True)
                case.counts[(i - 1) % K].copy_(prev[1], non_blocking=True)
    else:
        case.prepare(K, W, use_graph=not args.no_graph)
        graph_note = case.graph_note
        step = case.step
    log('launch mode:', graph_note)
    # W + 1 untimed steps (streaming mode: the first feed has no stage B), then exactly K timed steps, barrier + synchronize on
    # both sides, box gather inside the timed region, MAX over ranks
    tinfo = {}
    dt, all_b, all_c = fp.timed_steps(step, K, W + 1, case.results, case.counts, sync=torch.cuda.synchronize, info=tinfo)
    case.check_overflow()
    n_boxes = case.counts.float().mean().item()
    log('timed region: %d steps x %d frames in %.3f s' % (K, B, dt))

    out = None
    dtype_names = {'f32': 'f32', 'f16x2': 'f32 as f16 pairs (hi+lo, 22-bit significand; 3 f16 MFMA per product, f32 accumulate)',
                   'bf16x2': 'f32 as bf16 pairs (hi+lo, 16-bit significand; 3 bf16 MFMA per product, f32 accumulate)',
                   'f16': 'f16 products (hi halves of the f16 pairs: 11-bit inputs, 1 f16 MFMA per product), f32 accumulate, results stored as f16 pairs'}
    if rank == 0:
        value = world * K * B / dt
        out = {
            'metric': 'LiDAR frames/sec (160k pts, 0.1m voxels)', 'value': round(value, 3), 'unit': 'frames/s',
            'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': round(1000.0 * dt / K, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype_names[case.math], 'data': 'synthetic',
            'ranks_seen': tinfo.get('ranks_seen', 1), 'gather_ms': tinfo.get('gather_ms', 0.0),
            'config': {'workload': ('BASELINE configs[1]: %d-pt synthetic Waymo frames, 0.1 m voxels '
                                    '(grid 1504x1504x40), hard voxelize + MeanVFE + VoxelResBackBone8x + BaseBEVBackbone '
                                    '+ CenterHead + decode + rotated NMS, frames resident in HBM' % args.points) if args.sweeps == 1 else
                                   ('NOT the headline workload (--sweeps 2): BASELINE configs[4] shape, two merged sweeps per frame (2 x %d points), '
                                    'DynamicMeanVFE + 3-sweep model' % args.points),
                       'frames_per_step_per_gpu': B, 'ms_per_frame': round(1000.0 * dt / (K * B), 4), 'latency_ms_per_pass': round(1000.0 * dt / K, 4),
                       'like_for_like': 'leg ref_batch (8 frames per pass = BATCH_SIZE_PER_GPU of the reference config, centerpoint_1sweep.yaml:88) is the '
                                        'like-for-like batch; legs batch16 / batch32 are the headline configurations of rounds 1-3 / 4-6; value is at frames_per_step_per_gpu', 'parallelism': 'frame-parallel x%d' % world,
                       'concurrent_sub_passes': case.pipe.ways if case.pipe.splits(B) else 1,
                       'launch': graph_note, 'math': case.math, 'math_selected': case.math_selected, 'activation_peaks': case.activation_peaks,
                       'sparse_engine': args.sparse_engine,
                       'calibration': 'level capacities fitted (x1.5) on 4 frames of other seeds than the timed ones; overflow flag checked after the timed region',
                       'overlap': 'stage A (voxelize + index pyramid) of batch i+1 under stage B (convs, head, NMS) of batch i' if streamer is not None else 'none', 'weights': ('seeded synthetic set synth_detector(gain=preserve): variance-preserving, boxes depend on the frame and sit on its points (no checkpoints offline)'
                                   if args.weights == 'preserve' else 'NOT the headline weights (--weights default): the default initialisers of rounds 1-5, frame-independent boxes'),
                       'mean_boxes_per_frame': round(n_boxes, 1), 'boxes_per_frame_min_max': [int(case.counts.min().item()), int(case.counts.max().item())]},
        }

    # ---- roofline of the dominant kernel: HIP events around every conv launch, on the launch stream
    if rank == 0 and args.profile_frames > 0:
        kern, roof, _ = case.kernel_profile(args.profile_frames)
        if roof:
            roof['frames_per_step'] = case.PB               # frames of the profiled pass (one sub-pass of the step)
            out['roofline'] = roof
        out['kernels'] = kern
        log('per-kernel profile done')
        out['conv_ms_per_frame'] = round(sum(r['ms_per_step'] for r in kern) / case.PB, 4)        # (`kernels`: one pass of case.PB frames)
        out['profiled_pass_frames'] = case.PB

    # ---- auxiliary legs (single GPU only): what the headline does not show
    if rank == 0 and world == 1 and not args.no_aux and streamer is None:
        sec = args.aux_seconds
        try:
            out['stages'] = case.stage_profile()
            log('stages:', [(s['stage'], s['ms_per_step']) for s in out['stages']])
            # the north star's own fractions ("voxelize + sparse backbone at >= 60 % of the HBM roofline") inside `roofline`, so that
            # whoever keeps only that object can recompute them: algorithmic bytes (from the live counts), milliseconds
            # per step (each stage replayed as its own graph, HIP events on the launch stream), fraction of the 8 TB/s peak
            if isinstance(out.get('roofline'), dict):
                st = {r['stage']: r for r in out['stages']}
                hbm = {}
                for nme in ('voxelize', 'index', 'sparse_backbone'):
                    hbm[nme] = {'bytes': st[nme]['algorithmic_bytes'], 'ms': st[nme]['ms_per_step'], 'gbs': st[nme]['hbm_gbs'], 'frac': st[nme]['frac_of_hbm_peak']}
                vb_bytes = hbm['voxelize']['bytes'] + hbm['sparse_backbone']['bytes']
                vb_ms = hbm['voxelize']['ms'] + hbm['sparse_backbone']['ms']
                hbm['voxelize_plus_backbone'] = {'bytes': vb_bytes, 'ms': round(vb_ms, 4), 'gbs': round(vb_bytes / (vb_ms * 1e-3) / 1e9, 1),
                                                 'frac': round(vb_bytes / (vb_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
                # key order matters: the driver keeps only the head of this object, so the contract keys come first, the four
                # north-star fractions next, then what they are recomputed from (bytes, ms), the dominant kernel's detail last
                old = out['roofline']
                roof = {k: old[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic') if k in old}
                for nme in ('voxelize_plus_backbone', 'sparse_backbone', 'index', 'voxelize'):
                    roof['hbm_%s_frac' % nme] = hbm[nme]['frac']
                roof['hbm_peak_gbs'] = PEAK_HBM_GBS
                for nme in ('voxelize_plus_backbone', 'sparse_backbone', 'index', 'voxelize'):
                    for k in ('gbs', 'ms', 'bytes'):
                        roof['hbm_%s_%s' % (nme, k)] = hbm[nme][k]
                roof['dense_ms'] = st['dense']['ms_per_step']
                roof['post_ms'] = st['post']['ms_per_step']
                roof['frames_per_step'] = case.PB           # frames of the profiled pass (`kernels`, `stages`, the hbm_* entries): one sub-pass of the step
                for k, v in old.items():
                    roof.setdefault(k, v)
                out['roofline'] = roof
        except Exception as e:
            out['stages'] = {'error': str(e).split('\n')[0][:200]}
        # frames in pinned host memory: H2D of step i+1 on a copy stream under step i (double-buffered staging in HBM)
        hpool = torch.from_numpy(case.host_pool).pin_memory()
        stage = [torch.empty_like(case.static_in) for _ in range(2)]
        copy_stream = torch.cuda.Stream()
        ready = [torch.cuda.Event() for _ in range(2)]
        free = [torch.cuda.Event() for _ in range(2)]
        main_stream = torch.cuda.current_stream()
        for e in free:
            e.record(main_stream)
        state = {'next': None}

        def prefetch(i):
            o = (i * B) % case.n_distinct
            copy_stream.wait_event(free[i % 2])
            with torch.cuda.stream(copy_stream):
                stage[i % 2].copy_(hpool[o:o + B], non_blocking=True)
                ready[i % 2].record(copy_stream)
            state['next'] = i

        def step_h2d(i):
            if state['next'] != i:
                prefetch(i)
            main_stream.wait_event(ready[i % 2])
            case.static_in.copy_(stage[i % 2], non_blocking=True)
            free[i % 2].record(main_stream)
            prefetch(i + 1)
            case.run()
            kk = case.results.shape[0]
            case.results[i % kk].copy_(case.g_out, non_blocking=True)
            case.counts[i % kk].copy_(case.g_n, non_blocking=True)
        fps, ms, k = case.aux_leg(sec, step_h2d)
        out['with_h2d'] = {'value': round(fps, 2), 'unit': 'frames/s', 'ms_per_step': round(ms, 4), 'steps': k, 'frames_per_step': B,
                           'math': args.math, 'h2d_bytes_per_step': int(hpool[:B].numel() * 4),
                           'note': 'frames start in pinned host memory; the H2D copy of step i+1 (copy stream, double-buffered '
                                   'staging) overlaps step i; never the headline value'}
        log('with_h2d %.1f frames/s' % fps)
        del hpool, stage

        def leg(name, math, batch, lengths=None, mode='stacked', note='', sweeps=1):
            c = Case(args, dev, rank, math, batch, lengths, mode, seed_base=500, sweeps=sweeps)
            fps, ms, k = c.aux_leg(sec)
            rec = {'value': round(fps, 2), 'unit': 'frames/s', 'ms_per_step': round(ms, 4), 'steps': k, 'frames_per_step': c.B,
                   'math': math, 'dtype': dtype_names[math], 'launch': c.graph_note, 'mean_points_per_frame': round(c.mean_points), 'note': note}
            log('%s %.1f frames/s' % (name, fps))
            return c, rec
        c, rec = leg('fp32', 'f32', args.fp32_batch or B, note='every convolution on v_mfma_f32_16x16x4_f32: exact fp32 products and accumulation; same frames per pass as the headline unless --fp32-batch')
        if args.profile_frames > 0:
            kern, roof, _ = c.kernel_profile(args.profile_frames)
            rec['roofline'] = roof
            rec['kernels'] = kern[:6]
        out['fp32'] = rec
        # the precision-equivalent figure next to `value` (whose arithmetic carries 22 significant bits): promoted to top-level keys
        out['value_fp32'] = rec['value']
        out['roofline_fp32'] = rec.get('roofline')
        # ... and INSIDE `roofline`, the object the driver's record keeps: the reference-precision figure (exact fp32 on the fp32 matrix
        # cores, the reference's own arithmetic) at the same frames per pass, with the roofline fraction of ITS dominant kernel
        if isinstance(out.get('roofline'), dict):
            r32 = rec.get('roofline') or {}
            out['roofline'].update({'fp32_frames_per_s': rec['value'], 'fp32_frames_per_step': rec['frames_per_step'], 'fp32_ms_per_step': rec['ms_per_step'],
                                    'fp32_frac': r32.get('frac'), 'fp32_kernel': r32.get('kernel'), 'fp32_achieved': r32.get('achieved'),
                                    'fp32_peak': r32.get('peak'), 'fp32_unit': r32.get('unit'),
                                    'precision_note': 'value is measured in fp16 pairs (22-bit significand; per stage within 1.6x of the exact-fp32 engine\'s own '
                                                      'rounding error against float64, tests/test_gpu_full_parity.py::test_error_budget_against_float64); '
                                                      'fp32_frames_per_s is the same workload in exact fp32'})
        del c
        c, out['ref_batch'] = leg('ref_batch', args.math, REF_BATCH, note='BATCH_SIZE_PER_GPU of centerpoint_1sweep.yaml:88')
        del c
        if B != 16:
            c, out['batch16'] = leg('batch16', args.math, 16, note='16 frames per pass: the headline configuration of rounds 1-3 (round-over-round comparison)')
            del c
        if B != 32:
            c, out['batch32'] = leg('batch32', args.math, 32, note='32 frames per step: the headline configuration of rounds 4-6 (round-over-round comparison)')
            del c
        c, padded = leg('ragged/padded', args.math, B, (150000, 180000), 'padded',
                        note='frames of 150k-180k points padded with out-of-range rows to 180k-row slots: stacked dz_voxelize_to_level route')
        del c
        c, lst = leg('ragged/list', args.math, B, (150000, 180000), 'list',
                     note='slot j holds frames of its own length in 150k-180k: per-frame fused voxelizers on parallel streams')
        del c
        out['ragged'] = {'padded': padded, 'list': lst}
        if args.math == 'f16x2':
            c, rec = leg('f16', 'f16', B, note='opt-in fast mode on the same tensors: ONE fp16 MFMA per product (hi halves only) - plain-fp16 inputs, '
                                             'fp32 accumulation; not fp32-class (boxes within 3e-2 of the oracle on this workload, '
                                             'tests/test_gpu_f16.py) and never the headline value')
            if args.profile_frames > 0:
                kern, roof, _ = c.kernel_profile(args.profile_frames)
                rec['roofline'] = roof
                rec['kernels'] = kern[:6]
            out['f16'] = rec
            del c
        c, out['multisweep'] = leg('multisweep', args.math, min(B, 32), sweeps=2,          # (r02-r06e: 8 frames per step; 8 / 16 / 32 -> 686 / 752 / 778 frames/s)
                                   note='BASELINE configs[4] shape: two merged sweeps per frame (2 x %d points, 6 features incl. the time '
                                        'offset), DynamicMeanVFE + centerpoint_3sweeps backbone and head' % args.points)
        del c
        torch.cuda.empty_cache()
        if args.math != 'f32' and args.sparse_engine != 'gather':
            try:        # the previous default engine on the same frames (A/B on this box)
                c = Case(args, dev, rank, args.math, B, seed_base=500, engine='gather')
                fps, ms, k = c.aux_leg(sec)
                rec = {'value': round(fps, 2), 'unit': 'frames/s', 'ms_per_step': round(ms, 4), 'steps': k, 'frames_per_step': B, 'math': args.math,
                       'note': 'gather engine (one gathered row per (output row, tap) pair for every sparse convolution: sparse_conv_h.hip, '
                               'sparse_conv_w.h) - the default of rounds 1-3; the headline runs the submanifold convolutions of the 32 / 64 / '
                               '128-channel levels on the x-run engine (sparse_conv_x.hip)'}
                if args.profile_frames > 0:
                    kern, _, _ = c.kernel_profile(args.profile_frames)
                    rec['kernels'] = [r for r in kern if 'spconv' in r['kernel']]
                out['gather'] = rec
                log('gather engine %.1f frames/s' % fps)
                del c
            except Exception as e:
                out['gather'] = {'error': str(e).split('\n')[0][:200]}
            torch.cuda.empty_cache()
        if args.math != 'f32' and args.tiles_leg:
            try:
                c = Case(args, dev, rank, args.math, B, seed_base=500, engine='tiles')
                fps, ms, k = c.aux_leg(sec)
                rec = {'value': round(fps, 2), 'unit': 'frames/s', 'ms_per_step': round(ms, 4), 'steps': k, 'frames_per_step': B, 'math': args.math,
                       'note': 'opt-in tile-resident sparse engine (csrc/sparse_conv_t.hip: 512-row tiles, halo staged in LDS once per 16-channel '
                               'chunk, rows of every level in brick order) - parity-tested, slower than the gather engine inside the detector'}
                if args.profile_frames > 0:
                    kern, _, _ = c.kernel_profile(args.profile_frames)
                    rec['kernels'] = [r for r in kern if 'spconv' in r['kernel']]
                out['tiles'] = rec
                log('tiles %.1f frames/s' % fps)
                del c
            except Exception as e:
                out['tiles'] = {'error': str(e).split('\n')[0][:200]}
            torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        if not args.no_refine:
            # BASELINE configs[3]: the refiner's GRM / PRM at the reference's shapes, 1024 objects in chunks of 128 / 96
            try:
                import bench_refine
                r32 = bench_refine.measure(dev, 1024, 128, 2, 'f32', with_crop=True)
                r16 = bench_refine.measure(dev, 1024, 128, 2, 'f16x2', with_crop=False)
                out['refine'] = {
                    'workload': 'BASELINE configs[3]: 1024 objects; GRM 3 x 256 query + 4096 memory points, PRM 200 x 256 query + 200 x 48 memory points '
                                '(reference dataset defaults), random weights and inputs resident in HBM',
                    'grm_objects_per_s': {'f32': r32['grm_objects_per_s'], 'f16x2': r16['grm_objects_per_s']},
                    'prm_tracks_per_s': {'f32': r32['prm_objects_per_s'], 'f16x2': r16['prm_objects_per_s']},
                    'grm_tflops': {'f32': r32['grm_tflops'], 'f16x2': r16['grm_tflops']},
                    'prm_tflops': {'f32': r32['prm_tflops'], 'f16x2': r16['prm_tflops']},
                    'points_in_boxes_us': r32.get('points_in_boxes_us'),
                    'roofline': {'bound': 'mfma', 'kernel': 'k_mha_block', 'achieved': r32['mha_core_prm_tflops'], 'peak': PEAK_F32_MFMA_TFLOPS,
                                 'unit': 'TFLOP/s', 'frac': round(r32['mha_core_prm_tflops'] / PEAK_F32_MFMA_TFLOPS, 4),
                                 'avg_launch_us': r32['mha_core_prm_us'],
                                 'note': 'PRM cross-attention core: 96 tracks x 8 heads, 200 queries x 9600 keys x 32, exact fp32 on v_mfma_f32_16x16x4_f32; '
                                         'algorithmic FLOP = 4 * Lq * Lk * d per track'},
                    # the dominant kernels of the f16x2 refiner (PRM pass): algorithmic FLOP = 2 * rows * sum(cin * cout) of the layers a
                    # launch fuses, against the split-pair peak (three 16-bit MFMAs per product)
                    'roofline_f16x2': None if 'k_mlp_chain_prm_tflops' not in r16 else {
                        'bound': 'mfma', 'kernel': 'k_mlp_chain', 'achieved': r16['k_mlp_chain_prm_tflops'], 'peak': round(PEAK_F16_MFMA_TFLOPS / 3.0, 1),
                        'unit': 'TFLOP/s', 'frac': round(r16['k_mlp_chain_prm_tflops'] / (PEAK_F16_MFMA_TFLOPS / 3.0), 4),
                        'avg_launch_us': r16['k_mlp_chain_prm_us'], 'share_of_prm_pass': r16['k_mlp_chain_prm_share'],
                        'second': {'kernel': 'k_pointnet3', 'achieved': r16.get('k_pointnet3_prm_tflops'), 'avg_launch_us': r16.get('k_pointnet3_prm_us'),
                                   'frac': None if 'k_pointnet3_prm_tflops' not in r16 else round(r16['k_pointnet3_prm_tflops'] / (PEAK_F16_MFMA_TFLOPS / 3.0), 4),
                                   'share_of_prm_pass': r16.get('k_pointnet3_prm_share')},
                        'note': 'memory MLP 128 -> 512 -> 256 + K / V projections as one kernel (k_mlp_chain), PointNet 32 -> 128 -> 128 -> 512 + max as one '
                                'kernel (k_pointnet3); HIP events on the launch stream inside the PRM pass'},
                    'roofline_grm': None if 'xattn_folded_grm_gbs' not in r32 else {
                        'bound': 'hbm', 'kernel': 'k_xattn_fold', 'achieved': r32['xattn_folded_grm_gbs'], 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                        'frac': round(r32['xattn_folded_grm_gbs'] / PEAK_HBM_GBS, 4), 'avg_launch_us': r32['xattn_folded_grm_us'],
                        'note': 'GRM cross-attention, 128 objects x 3 queries x 4096 memory rows x 256 channels x 8 heads with the key / value projections folded '
                                'into the queries (fold + attend + merge launches together): algorithmic bytes = the memory rows, read once'}}
                log('refine GRM %.0f / %.0f objects/s, PRM %.0f / %.0f tracks/s (f32 / f16x2)' % (
                    r32['grm_objects_per_s'], r16['grm_objects_per_s'], r32['prm_objects_per_s'], r16['prm_objects_per_s']))
            except Exception as e:
                out['refine'] = {'error': str(e).split('\n')[0][:200]}
            torch.cuda.empty_cache()
        if not args.no_pdv:
            try:
                import bench_pdv
                out['pdv'] = bench_pdv.measure(dev, args.points, 8, 'f32')
                p16 = bench_pdv.measure(dev, args.points, 8, 'f16x2')
                out['pdv']['f16x2'] = {k: p16[k] for k in ('first_stage_ms', 'second_stage_ms', 'rois_per_s', 'frames_per_s', 'rois', 'rois_proposed', 'rois_nonempty_frac')}
                # the same modules over 8 frames per pass (one batch_dict: every kernel of both stages launched once for all frames / RoIs)
                p16b = bench_pdv.measure(dev, args.points, 4, 'f16x2', batch=8)
                out['pdv']['f16x2_batch8'] = {k: p16b[k] for k in ('frames_per_pass', 'first_stage_ms', 'second_stage_ms', 'rois_per_s', 'frames_per_s', 'rois', 'rois_proposed', 'rois_nonempty_frac')}
                # FramePipeline.two_stage: first stage batched and sync-free, roi_head once over the batch
                keys = ('frames_per_pass', 'first_stage_ms', 'second_stage_ms', 'rois_per_s', 'frames_per_s', 'rois', 'rois_proposed', 'rois_nonempty_frac')
                for nb in (8, 16):
                    pp = bench_pdv.measure(dev, args.points, 4, 'f16x2', batch=nb, pipeline=True)
                    out['pdv']['f16x2_pipeline_batch%d' % nb] = {k: pp[k] for k in keys}
                out['pdv']['frames_per_s_best'] = max(out['pdv'][k]['frames_per_s'] for k in ('f16x2_batch8', 'f16x2_pipeline_batch8', 'f16x2_pipeline_batch16'))
                log('pdv first stage %.2f ms, second stage %.2f ms (%d RoIs); f16x2: %.2f + %.2f ms; pipeline x16: %.1f frames/s' % (
                    out['pdv']['first_stage_ms'], out['pdv']['second_stage_ms'], out['pdv']['rois'], p16['first_stage_ms'], p16['second_stage_ms'],
                    out['pdv']['f16x2_pipeline_batch16']['frames_per_s']))
            except Exception as e:
                out['pdv'] = {'error': str(e).split('\n')[0][:200]}
            torch.cuda.empty_cache()

    # ---- CPU baseline: the oracle (reference-semantics restatement) on this host's cores, bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from tests.util import cpu_state_dict, oracle_detect       # oracle = checker/baseline only
        cores = min(usable_cores(), 32)
        torch.set_num_threads(cores)
        sd = cpu_state_dict(case.model)
        pts = case.host_frames
        from oracle.voxelize import mask_points_by_range

        def run_oracle(budget, max_frames, warm):
            """-> (per-frame seconds list, per-stage seconds summed) within `budget` seconds of CPU time"""
            times, stage, spent, i = [], {}, 0.0, 0
            for _ in range(warm):                                   # untimed warm-up frames (allocator, thread pool)
                p = pts[0]
                oracle_detect(sd, p[mask_points_by_range(p, case.info.point_cloud_range)], case.info)
            while spent < budget and i < max_frames:
                p = pts[i % case.n_distinct]
                p = p[mask_points_by_range(p, case.info.point_cloud_range)]
                t1 = time.perf_counter()
                oracle_detect(sd, p, case.info, times=stage)
                times.append(time.perf_counter() - t1)
                spent += times[-1]
                i += 1
            return times, stage
        # all host cores and one thread, warm-up, median over the frames that fit the time budget, per-stage split
        times, stage = run_oracle(args.cpu_baseline_seconds, 12, 1)
        med = float(np.median(times))
        torch.set_num_threads(1)
        t1x, stage1 = run_oracle(args.cpu_baseline_seconds * 0.6, 3, 0)
        torch.set_num_threads(cores)
        n = len(times)
        out['cpu_baseline'] = {'value': round(1.0 / med, 4), 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
                               'sample': 'median of %d frames (after 1 warm-up frame) of the same 160k-pt workload through oracle/ (numpy + CPU torch '
                                         'restatement of the spconv/PyTorch path; spconv itself is not installable), %.1f s of CPU time' % (n, sum(times)),
                               'frames': n, 'mean_value': round(n / sum(times), 4),
                               'stages_ms_per_frame': {k: round(1000.0 * v / n, 1) for k, v in stage.items()},
                               'one_thread': {'value': round(len(t1x) / sum(t1x), 4), 'cores': 1, 'frames': len(t1x),
                                              'stages_ms_per_frame': {k: round(1000.0 * v / len(t1x), 1) for k, v in stage1.items()}}}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
