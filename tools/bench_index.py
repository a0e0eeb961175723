#!/usr/bin/env python
"""Voxelize + index-pyramid chain of the headline workload on its own (no convolutions running beside it): wall time per step from
HIP events, and - under `rocprofv3 --kernel-trace` - the per-kernel durations of the chain without the overlap of the detector's
side stream.  usage: python tools/bench_index.py [--reps 20] [--batch 16]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--points', type=int, default=160000)
    a = ap.parse_args()
    sys.argv = [sys.argv[0], '--batch', str(a.batch), '--points', str(a.points)]
    args = bench.parse()
    dev = torch.device('cuda:0')
    case = bench.Case(args, dev, 0, 'f16x2', a.batch)
    pipe = case.pipe
    from detzero_amd import ops
    from detzero_amd.centerpoint import _StackedFrames
    frames = _StackedFrames(case.static_in.contiguous())

    def chain():
        vox = pipe.voxelize_stage(frames)
        return vox, pipe.pyramid_stage(vox, a.batch, overlap=False)
    for _ in range(3):
        vox, pyr = chain()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tv = tp = 0.0
    for _ in range(a.reps):
        ev[0].record()
        vox = pipe.voxelize_stage(frames)
        ev[1].record()
        pyr = pipe.pyramid_stage(vox, a.batch, overlap=False)
        ev[2].record()
        torch.cuda.synchronize()
        tv += ev[0].elapsed_time(ev[1]) / a.reps
        tp += ev[1].elapsed_time(ev[2]) / a.reps
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g):
            vox, pyr = chain()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    rows = [st[2].num_active() for st in pyr['steps']]
    pairs = []
    for nbr_d, nbr_s, lvl, _ in pyr['steps']:
        m = lvl.num_active()
        pairs.append([None if t is None else ops.table_pairs(t, m) for t in (nbr_d, nbr_s)])
    print(json.dumps({'batch': a.batch, 'points': a.points, 'eager_voxelize_ms': round(tv, 4), 'eager_pyramid_ms': round(tp, 4),
                      'graph_chain_ms': round(e0.elapsed_time(e1) / a.reps, 4), 'rows': rows, 'pairs': pairs,
                      'caps': [st[2].cap for st in pyr['steps']]}))


if __name__ == '__main__':
    main()
