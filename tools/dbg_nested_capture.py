#!/usr/bin/env python
"""Reproducer (ROCm 7.2, MI355X): hipStreamEndCapture segfaults when a stream that joined the capture through a FORKED branch forks
again (two concurrent FramePipeline sub-passes, each with its index pyramid on a side stream, inside ONE capture).
    python tools/dbg_nested_capture.py nooverlap   -> eager ok / capture ok / replay ok     (sub-passes without inner forks)
    python tools/dbg_nested_capture.py plain       -> eager ok / Segmentation fault in capture_end
This is why FramePipeline._call_split runs its sub-passes with overlap=False inside a capture."""
import faulthandler
import os
import sys

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from detzero_amd.centerpoint import FramePipeline, _StackedFrames, synth_detector  # noqa: E402
from detzero_amd.synth import VOXEL_SIZE_02, synth_waymo_frame  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'plain'
dev = torch.device('cuda', 0)
model, cfg, info = synth_detector(VOXEL_SIZE_02, seed=0)
model = model.to(dev)
frames = [torch.from_numpy(synth_waymo_frame(80 + i, 20000)).to(dev) for i in range(4)]
inp = torch.stack(frames)
pipes = [FramePipeline(model, info, math='f16x2', ways=1) for _ in range(2)]
for k, p in enumerate(pipes):
    p.side_key = 1 + k
    p.calibrate(frames[:2])
streams = [torch.cuda.Stream() for _ in range(2)]


def run():
    main = torch.cuda.current_stream()
    outs = []
    for k in range(2):
        part = _StackedFrames(inp[2 * k:2 * k + 2])
        streams[k].wait_stream(main)
        with torch.cuda.stream(streams[k]):
            outs.append(pipes[k].infer(pipes[k].prepare(part, overlap=(mode != 'nooverlap'))))
    for st in streams:
        main.wait_stream(st)
    return outs


for _ in range(2):
    run()
torch.cuda.synchronize()
print(mode, 'eager ok', flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    o = run()
print(mode, 'capture ok', flush=True)
g.replay()
torch.cuda.synchronize()
print(mode, 'replay ok', flush=True)
