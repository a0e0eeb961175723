import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detzero_amd import ops
from tests.test_gpu_xrun import _level, _t, K3, S1, P1
dev = torch.device('cuda', 0)
rng = np.random.default_rng(1)
for shape, dens, batch in (([3, 20, 33], (0.08,), 1), ([6, 36, 50], (0.3, 0.35, 0.25), 2), ([8, 200, 200], (0.2,), 2)):
    lvl, coords = _level(rng, batch, shape, dens, dev)
    m = coords.shape[0]
    for c in (32, 64, 128):
        xt = ops.build_windows(lvl.neighbors_to(lvl, K3, S1, P1, packed=True), lvl, c)
        win, tr = xt.xwin[:2]
        nt = (lvl.cap + tr - 1) // tr
        x = ops.pair16_from_f32(torch.randn((lvl.cap, c), device=dev), c, 1)
        w = ops.pack_weight_split(torch.randn((27, c, c), device=dev) * 0.05, 1)
        one, zero = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        plain = lvl.neighbors_to(lvl, K3, S1, P1)
        ref = ops.pair16_to_f32(ops.spconv_forward(x, plain, lvl, w, one, zero, None, relu=False, math=1), 1)[:m]
        for rep in range(3):
            out = ops.pair16_to_f32(ops.spconv_forward(x, xt, lvl, w, one, zero, None, relu=False, math=1), 1)[:m]
            torch.cuda.synchronize()
            print('rows', m, 'c', c, 'rep', rep, 'maxdiff', float((out - ref).abs().max()), 'queue', win[nt * 6:nt * 6 + 9].tolist(), flush=True)
