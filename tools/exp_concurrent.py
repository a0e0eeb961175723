#!/usr/bin/env python
"""Experiment (round 6): K concurrent sub-passes of B / K frames on K streams against one pass of B frames, frames/s of the whole
detector step, hipGraph replay.  Persistent kernels leave CUs idle during every launch's ramp and tail (busy CUs 0.84-0.96,
profiles/r06a_clock_table.txt); a second, independent pass fills them.
    python tools/exp_concurrent.py [--batch 32] [--ways 1,2,4] [--steps 60]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--ways', default='1,2,4')
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--points', type=int, default=160000)
    args = ap.parse_args()
    from detzero_amd.centerpoint import FramePipeline, synth_detector
    from detzero_amd.synth import VOXEL_SIZE_01, synth_waymo_frame
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    model, cfg, info = synth_detector(VOXEL_SIZE_01, seed=0)
    model = model.to(dev)
    B = args.batch
    frames = np.stack([synth_waymo_frame(500 + i, args.points) for i in range(2 * B + 1)])
    pool = torch.from_numpy(frames).to(dev)
    cal = [torch.from_numpy(synth_waymo_frame(7000 + i, args.points)).to(dev) for i in range(4)]
    if os.environ.get('EXP_SELECT'):
        from detzero_amd.centerpoint import select_math
        print('select_math', select_math(model, info, cal[:2])[0], model.prescale)
    if os.environ.get('EXP_CAPTURE'):
        pipe = FramePipeline(model, info, math='f16x2', ways=2)
        pipe.calibrate(cal)
        static_in = pool[:B].clone()
        for _ in range(3):
            pipe(static_in)
        torch.cuda.synchronize()
        cp = pipe.capture(static_in)
        for i in range(5):
            cp.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            o = (i * B) % (B + 1)
            static_in.copy_(pool[o:o + B], non_blocking=True)
            cp.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('pipe.capture ways=2: %8.1f frames/s  %7.3f ms per step, %d branches' % (args.steps * B / dt, 1000 * dt / args.steps, cp.branches), flush=True)
        del cp, pipe
    for ways in [int(w) for w in args.ways.split(',')]:
        nb = B // ways
        subs = []
        for k in range(ways):
            pipe = FramePipeline(model, info, math='f16x2', ways=1)
            pipe.side_key = 1 + k if os.environ.get('EXP_SIDE_KEYS') else 0
            pipe.calibrate(cal)
            static_in = pool[k * nb:(k + 1) * nb].clone()
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for _ in range(3):
                    out = pipe(static_in)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                out = pipe(static_in)
            torch.cuda.synchronize()
            subs.append((pipe, static_in, stream, g, out))
        main_s = torch.cuda.current_stream()

        def step(i):
            for k, (pipe, static_in, stream, g, out) in enumerate(subs):
                stream.wait_stream(main_s)
                with torch.cuda.stream(stream):
                    o = (i * B + k * nb) % 4
                    static_in.copy_(pool[o + k * nb:o + (k + 1) * nb] if o + (k + 1) * nb <= pool.shape[0] else pool[k * nb:(k + 1) * nb], non_blocking=True)
                    g.replay()
            for _, _, stream, _, _ in subs:
                main_s.wait_stream(stream)
        for i in range(5):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ov = any(p.overflow_seen() for p, *_ in subs)
        print('ways %d x %2d frames: %8.1f frames/s  %7.3f ms per %d-frame step  boxes %s overflow %s' % (
            ways, nb, args.steps * B / dt, 1000 * dt / args.steps, B, [int(s[4][1].sum().item()) for s in subs], ov), flush=True)
        del subs
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
