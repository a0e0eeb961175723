"""Timing of the object-feature kernels on one GPU (development / DESIGN.md numbers): a GRM batch and a PRM batch of
synthetic object tracks; kernel time with HIP events around the C-ABI calls, host packing / index drawing separately."""
import argparse
import random
import sys
import time
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detzero_amd import object_features as of          # noqa: E402
from detzero_amd.synth import synth_object_track       # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / reps, (time.perf_counter() - t0) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--objects', type=int, default=128)
    ap.add_argument('--frames', type=int, default=100)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    tracks = [synth_object_track(100 + i, a.frames, 'Vehicle', 20, 400) for i in range(a.objects)]
    t0 = time.perf_counter()
    packed = of.PackedTracks(tracks, dev)
    torch.cuda.synchronize()
    print('pack + H2D of %d objects, %d boxes, %d points: %.1f ms' % (a.objects, sum(packed.box_num), packed.pts.shape[0], (time.perf_counter() - t0) * 1e3))
    random.seed(0)
    dd = of.DeviceDraw(seed=1)
    for name, fn in (('GRM', lambda: of.grm_features(packed)), ('PRM', lambda: of.prm_features(packed)),
                     ('GRM device draw', lambda: of.grm_features(packed, rng=dd)), ('PRM device draw', lambda: of.prm_features(packed, rng=dd))):
        out, dev_ms, wall_ms = timed(fn)
        nbytes = sum(v.numel() * v.element_size() for v in out.values() if torch.is_tensor(v))
        print('%s features: %.1f MB out, wall %.1f ms per batch (host index drawing + H2D + kernels), stream %.2f ms' % (name, nbytes / 1e6, wall_ms, dev_ms))
    # kernels alone: re-launch with fixed index lists
    lib = of.L.load()
    q_idx, m_idx = of.prm_selection(packed)
    codes = np.asarray([0, 1, 2, 3], dtype=np.int32)
    b = packed.batch
    bufs = [torch.empty(s, dtype=torch.float32, device=dev) for s in ((b, 200, 256, 32), (b, 200, 48, 32), (b, 200, 7), (b, 200))]
    init = torch.empty((b, 7), dtype=torch.float64, device=dev)
    scratch = torch.empty((b * 200 * 27 + 2 * b,), dtype=torch.float64, device=dev)
    dq, dm = torch.from_numpy(q_idx).to(dev), torch.from_numpy(m_idx).to(dev)

    def launch():
        rc = lib.dz_prm_encode_points(of.L.ptr(packed.pts), of.L.ptr(packed.box_offsets), of.L.ptr(packed.traj), of.L.ptr(packed.score),
                                      of.L.ptr(packed.obj_box_offsets), of.L.ptr(packed.obj_cls), of.L.ptr(dq), of.L.ptr(dm), 256, 48, b, 200,
                                      codes.ctypes.data, 4, *[of.L.ptr(x) for x in bufs], of.L.ptr(init), of.L.ptr(scratch), of.L.stream())
        of.L.check(rc, 'dz_prm_encode_points')
    _, ms, _ = timed(launch, reps=20)
    out_bytes = sum(x.numel() * 4 for x in bufs)
    print('dz_prm_encode_points alone: %.3f ms for %.1f MB written = %.2f TB/s' % (ms, out_bytes / 1e6, out_bytes / ms / 1e9))


if __name__ == '__main__':
    main()
