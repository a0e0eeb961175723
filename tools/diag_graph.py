"""Diagnostic: eager vs HIP-graph replay of the frame pipeline (prints progress, syncs after each step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detzero_amd.centerpoint import FramePipeline, synth_detector
from detzero_amd.synth import VOXEL_SIZE_02, VOXEL_SIZE_01
from tests.util import masked_frame

def P(*a):
    print(*a, flush=True)

dev = torch.device('cuda', 0)
vs = VOXEL_SIZE_01 if '--full' in sys.argv else VOXEL_SIZE_02
npts = 160000 if '--full' in sys.argv else 20000
model, cfg, info = synth_detector(vs, seed=2)
model = model.to(dev)
pipe = FramePipeline(model, info)
frames = [torch.from_numpy(masked_frame(s, npts)).to(dev) for s in (3, 4, 5)]
outs = []
for i, f in enumerate(frames):
    o, n = pipe(f); torch.cuda.synchronize()
    outs.append((o.clone(), int(n.item())))
    P('eager frame', i, 'points', f.shape[0], 'boxes', outs[-1][1])
nmax = max(f.shape[0] for f in frames)
static_in = torch.zeros((nmax, 5), device=dev)
def load(f):
    static_in.zero_(); static_in[:, 0] = 1e6; static_in[:f.shape[0]] = f
load(frames[0])
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        pipe(static_in)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize(); P('warm ok')
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    g_out, g_n = pipe(static_in)
torch.cuda.synchronize(); P('captured')
for rep in range(3):
    for i, f in enumerate(frames):
        load(f); torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        n = int(g_n.item())
        same = (n == outs[i][1]) and torch.equal(g_out[:n], outs[i][0][:n])
        P('replay', rep, 'frame', i, 'boxes', n, 'equal_to_eager', same)
P('graph diag done')
