#!/usr/bin/env python
"""Timing of the refiner's fused kernels alone at the PRM / GRM chunk shapes (development tool; with a -DDZ_CHAIN_DIAG build the results
are garbage and only the times mean something).    python tools/bench_chain.py [--math f16x2]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1000.0 * e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--math', default='f16x2')
    args = ap.parse_args()
    from detzero_amd import ops
    dev = torch.device('cuda', 0)
    mid = ops.math_id(args.math)
    g = torch.Generator().manual_seed(0)

    def w(ci, co):
        return ops.pack_weight_split((torch.randn((ci, co), generator=g) / ci ** 0.5).to(dev), mid)

    def vec(n, fill=None):
        return (torch.rand(n, generator=g) + 0.5).to(dev) if fill is None else torch.full((n,), fill, device=dev)
    for name, groups, length, kv in (('PRM memory chain', 96, 9600, True), ('GRM memory chain', 128, 4096, False)):
        rows = groups * length
        x = ops.pair16_from_f32(torch.randn((rows, 128), generator=g).clamp_(min=0).to(dev), 128, mid)
        la, lb = (w(128, 512), vec(512), vec(512, 0.1)), (w(512, 256), vec(256), vec(256, 0.1))
        gs = torch.randn((groups, 512), generator=g).to(dev)
        kvw = (w(256, 256), vec(256, 0.0), w(256, 256), vec(256, 0.0)) if kv else None
        us = timed(lambda: ops.mlp_chain(x, la, lb, gs, length, mid, kv=kvw))
        macs = rows * (128 * 512 + 512 * 256 + (2 * 256 * 256 if kv else 0))
        print('%-18s %8d rows  %8.1f us  %6.1f TF/s algorithmic (%.0f %% of the pair16 peak)' % (name, rows, us, 2e-6 * macs / us, 2e-6 * macs / us / 838.9 * 100))
        del x
    for name, groups, length, c3, cin in (('PRM query encoder', 96 * 200, 256, 256, 32), ('PRM memory encoder', 96, 9600, 256, 32), ('GRM memory encoder', 128, 4096, 512, 32)):
        rows = groups * length
        x = torch.randn((rows, cin), generator=g).to(dev)
        trip = [(w(32, 128), vec(128), vec(128, 0.1)), (w(128, 128), vec(128), vec(128, 0.1)), (w(128, c3), vec(c3), vec(c3, 0.1))]
        us = timed(lambda: ops.pointnet3(x, trip, length, mid, want_tap=length > 256, x_f32=True))
        macs = rows * (32 * 128 + 128 * 128 + 128 * c3)
        print('%-18s %8d rows  %8.1f us  %6.1f TF/s algorithmic (%.0f %% of the pair16 peak)' % (name, rows, us, 2e-6 * macs / us, 2e-6 * macs / us / 838.9 * 100))
        del x


if __name__ == '__main__':
    main()
