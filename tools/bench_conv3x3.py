"""One BEV 3x3 layer (16 x 188 x 188, 128 -> 128 channels, pair16) timed alone: us per launch and algorithmic TF/s.
Development tool: with a -DDZ_C3_DIAG build, DZ_TUNE_C3_DIAG=<bits> removes one effect at a time (results are garbage)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from detzero_amd import ops                      # noqa: E402
from detzero_amd.det_modules import conv_layer   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--hw', type=int, default=188)
    ap.add_argument('--cin', type=int, default=128)
    ap.add_argument('--cout', type=int, default=128)
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--math', type=int, default=1, help='1 = f16x2, 2 = bf16x2, 3 = f16 (single product)')
    ap.add_argument('--data', default='randn', choices=['randn', 'relu', 'zero', 'const'])
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(0)
    h = w = a.hw
    x = torch.zeros(a.batch, h + 2, w + 2, a.cin)
    x[:, 1:-1, 1:-1] = torch.randn(a.batch, h, w, a.cin, generator=g)
    wr = torch.randn(9, a.cin, a.cout, generator=g) * 0.05
    if a.data == 'relu':
        x = x.clamp_min(0)
    elif a.data == 'zero':
        x, wr = x * 0, wr * 0
    elif a.data == 'const':
        x, wr = (x != 0).float(), wr * 0 + 0.5
    xp = ops.pair16_from_f32(x.to(dev), math=1)
    wt = ops.pack_weight_split(wr.to(dev), 1)
    scale = torch.ones(wt.shape[-2], device=dev)
    shift = torch.zeros(wt.shape[-2], device=dev)
    y = torch.zeros(a.batch, h + 2, w + 2, a.cout, device=dev)

    def run():
        conv_layer(xp, (h + 2, w + 2), wt, scale, shift, True, y, (h + 2, w + 2), cin=a.cin, in_cstride=a.cin, ksize=3,
                   stride=1, in_off=0, out_cstride=a.cout, out_d=(1, 1), ho=h, wo=w, batch=a.batch, math=a.math)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    fl = 2.0 * a.batch * h * w * 9 * a.cin * a.cout
    print('conv3x3 %dx%dx%d %d->%d data=%s diag=%s  %.1f us  %.1f TF/s algorithmic' % (a.batch, h, w, a.cin, a.cout, a.data,
          os.environ.get('DZ_TUNE_C3_DIAG', '0'), us, fl / us * 1e-6))


if __name__ == '__main__':
    main()
