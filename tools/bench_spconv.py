#!/usr/bin/env python
"""Per-layer timing of the sparse 3-D backbone on one MI355X (development tool, not the headline bench).

    python tools/bench_spconv.py [--batch 4] [--reps 20] [--points 160000]

Builds the sparse levels of a batch of synthetic frames once, then times every distinct sparse conv of
VoxelResBackBone8x separately (HIP events on the launch stream) and prints rows, rulebook pairs, mean valid
taps per row, taps with at least one pair per 16-/64-/128-row tile, algorithmic TF/s and the "dense-tap"
TF/s (what the matrix pipe really executes with tile-level tap skipping).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--points', type=int, default=160000)
    ap.add_argument('--math', default='f32')
    ap.add_argument('--zero', action='store_true', help='time the layers on all-zero activations and weights (matrix-pipe power test)')
    ap.add_argument('--only', default='', help='comma list of cin-cout pairs to time (default: every layer)')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    from detzero_amd import ops
    from detzero_amd.centerpoint import FramePipeline, synth_detector
    from detzero_amd.synth import VOXEL_SIZE_01, synth_waymo_frame
    model, cfg, info = synth_detector(VOXEL_SIZE_01, seed=0)
    model = model.to(dev)
    pipe = FramePipeline(model, info, math=args.math)
    mm = ops.math_id(args.math)
    frames = [torch.from_numpy(synth_waymo_frame(i, args.points)).to(dev) for i in range(args.batch)]
    pipe.calibrate(frames[:4])
    feats, coords, d_n = pipe._voxelize(frames)

    calls = []
    real = ops.spconv_forward

    def spy(f, nbr, out_level, w, scale, shift, residual=None, relu=True, out=None, in_level=None, math=0, cout=None):
        calls.append((f, nbr, out_level, w, scale, shift, residual, relu, in_level))
        return real(f, nbr, out_level, w, scale, shift, residual, relu, out, in_level, math, cout)
    ops.spconv_forward = spy
    import detzero_amd.det_modules as dm
    dm.ops.spconv_forward = spy
    model.backbone3d.run_pyramid(model.backbone3d.build_pyramid(feats, coords, args.batch, d_n, caps=[c * args.batch for c in pipe.level_caps]))
    ops.spconv_forward = real
    dm.ops.spconv_forward = real
    torch.cuda.synchronize()

    seen = {}
    total = 0.0
    for (f, nbr, lvl, w, scale, shift, residual, relu, in_level) in calls:
        kvol = w.shape[0]
        variant = 'x' if getattr(nbr, 'xwin', None) is not None and w.shape[1] == w.shape[2] else 'g'
        cin, cout = (w.shape[2], scale.shape[0]) if mm else (w.shape[1], w.shape[2])
        key = (kvol, cin, cout, lvl.cap, residual is not None, id(nbr))
        if args.only and '%d-%d' % (cin, cout) not in args.only.split(','):
            continue
        m = lvl.num_active()
        if key not in seen:
            valid = ops.unpack_table(nbr)[:, :m] >= 0
            pairs = int(valid.sum().item())

            def tile_taps(bm):
                pad = (-m) % bm
                v = torch.cat([valid, valid.new_zeros((kvol, pad))], dim=1).view(kvol, -1, bm).any(dim=2)
                return float(v.sum().item()) / v.shape[1]
            stats = (pairs, tile_taps(16), tile_taps(32), tile_taps(64), tile_taps(128))
            out = torch.empty((lvl.cap, cout), dtype=torch.float32, device=dev)
            if args.zero:
                f, w = torch.zeros_like(f), torch.zeros_like(w)
                residual = torch.zeros_like(residual) if residual is not None else None
            for _ in range(3):
                real(f, nbr, lvl, w, scale, shift, residual, relu, out, in_level, mm)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                real(f, nbr, lvl, w, scale, shift, residual, relu, out, in_level, mm)
            e1.record()
            torch.cuda.synchronize()
            us = 1000.0 * e0.elapsed_time(e1) / args.reps
            if os.environ.get('DZ_TUNE_X_DIAG') and variant == 'x':
                import ctypes
                from detzero_amd import lib as L
                try:
                    L.load().dz_spconv_x_debug_dump()
                except AttributeError:
                    pass
            seen[key] = (us, stats)
        us, (pairs, t16, t32, t64, t128) = seen[key]
        halo = ''
        if getattr(nbr, 'tiles', None) is not None:
            nh = nbr.tiles[1][:(m + 511) // 512].float()
            halo = '  tiles: halo/rows %.2f max %d' % (float(nh.sum().item()) / max(m, 1), int(nh.max().item()))
        total += us
        flop = 2.0 * pairs * cin * cout
        print(variant + ' k%-2d %3d->%-3d rows %8d pairs/row %5.2f taps/tile[16|32|64|128] %5.2f %5.2f %5.2f %5.2f  %8.1f us  alg %6.2f TF/s  '
              'dense64 %6.2f TF/s%s' % (kvol, cin, cout, m, pairs / max(m, 1), t16, t32, t64, t128, us, flop / us / 1e6,
                                        2.0 * m * t64 * cin * cout / us / 1e6, ('  +res' if residual is not None else '') + halo))
    print('sum over the %d sparse convs: %.1f us per step (%.1f us per frame)' % (len(calls), total, total / args.batch))


if __name__ == '__main__':
    main()
