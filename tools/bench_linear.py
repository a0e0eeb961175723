#!/usr/bin/env python
"""Per-shape timing of dz_linear_forward_split on the refiner's layer shapes (development tool).

    python tools/bench_linear.py [--math f16x2]

rows x cin -> cout as they occur in one chunk of GRM (128 objects x 4096 / 3 x 256 points) and PRM (96 tracks x 200 x 256 /
200 x 48 points); prints time, HBM GB/s of the compulsory bytes (pair16 = 4 bytes per value) and algorithmic TF/s.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--math', default='f16x2')
    ap.add_argument('--reps', type=int, default=5)
    args = ap.parse_args()
    from detzero_amd import ops
    dev = torch.device('cuda', 0)
    mid = ops.math_id(args.math)
    shapes = [('GRM mem 1', 524288, 32, 128, 0), ('GRM mem 2', 524288, 128, 128, 0), ('GRM mem 3 +max', 524288, 128, 512, 4096),
              ('GRM mlp 1', 524288, 128, 512, 0), ('GRM mlp 2', 524288, 512, 256, 0), ('GRM k/v', 524288, 256, 256, 0),
              ('PRM q 1', 4915200, 32, 128, 0), ('PRM q 2', 4915200, 128, 128, 0), ('PRM q 3 +max', 4915200, 128, 256, 256),
              ('PRM mem 1', 921600, 32, 128, 0), ('PRM mem 2', 921600, 128, 128, 0), ('PRM mem 3 +max', 921600, 128, 256, 9600),
              ('PRM mlp 1', 921600, 128, 512, 0), ('PRM mlp 2', 921600, 512, 256, 0), ('PRM k/v', 921600, 256, 256, 0)]
    g = torch.Generator().manual_seed(0)
    for name, rows, cin, cout, gmax in shapes:
        x = ops.pair16_from_f32(torch.randn((rows, cin), generator=g).to(dev), cin, mid)
        w = ops.pack_weight_split((torch.randn((cin, cout), generator=g) / cin ** 0.5).to(dev), mid)
        sc = torch.ones(w.shape[0], device=dev)
        sh = torch.zeros(w.shape[0], device=dev)

        def run():
            if gmax:
                return ops.linear_split(x, w, sc, sh, True, cout, mid, group_rows=gmax, group_max=True)
            return ops.linear_split(x, w, sc, sh, True, cout, mid)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = 1000.0 * e0.elapsed_time(e1) / args.reps
        nbytes = 4.0 * rows * cin + (0 if gmax else 4.0 * rows * cout)
        print('%-16s %8d x %3d -> %3d  %8.1f us  %7.1f GB/s  %6.1f TF/s' % (name, rows, cin, cout, us, nbytes / us / 1e3, 2.0 * rows * cin * cout / us / 1e6))
        del x, w


if __name__ == '__main__':
    main()
