import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from detzero_amd.centerpoint import FramePipeline, synth_detector
from detzero_amd.synth import VOXEL_SIZE_02, synth_waymo_frame
dev = torch.device('cuda', 0)
model, cfg, info = synth_detector(VOXEL_SIZE_02, seed=0)
model = model.to(dev)
frames = [torch.from_numpy(synth_waymo_frame(i, 20000)).to(dev) for i in range(4)]
p1 = FramePipeline(model, info, math='f16x2', chains=1)
o1, n1 = p1(frames)
torch.cuda.synchronize(); print('chains=1 ok', n1.tolist(), flush=True)
p2 = FramePipeline(model, info, math='f16x2', chains=2)
o2, n2 = p2(frames)
torch.cuda.synchronize(); print('chains=2 ok', n2.tolist(), flush=True)
print('equal', torch.equal(o1, o2))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): p2(frames)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    go, gn = p2(frames)
g.replay(); torch.cuda.synchronize()
print('graph ok', torch.equal(go, o1))
