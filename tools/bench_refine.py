#!/usr/bin/env python
"""Secondary kernel set (BASELINE.json configs[3]): refining module throughput on one MI355X.

    python tools/bench_refine.py [--objects 1024] [--steps 5]

GRM: per object 3 proposals x 256 query points + 4096 memory points (reference defaults,
refining/tools/cfgs/ref_dataset_cfgs/waymo_grm_dataset.yaml); PRM: per track 200 boxes x 256 query points +
200 x 48 memory points.  Random weights (synth_state_dict), random inputs resident in HBM, fp32 MFMA.
Also times the object crop mask (points_in_boxes_gpu_v2) on a 180k-point frame x 128 boxes.
Prints one JSON line with objects/s and the attention-core / GEMM rates.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def timed(fn, steps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def measure(dev, objects=1024, chunk=128, steps=3, math='f32', with_crop=True):
    """Refiner throughput on `dev` -> dict (also the `refine` leg of bench.py: BASELINE configs[3])."""
    from detzero_amd import ops
    from detzero_amd.refine_modules import GeometryTransformer, PositionTransformer
    from detzero_amd.synth import synth_boxes, synth_state_dict, synth_waymo_frame
    if os.path.join(ROOT, 'tests') not in sys.path:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_refine import GCFG, PCFG
    out = {'metric': 'refiner objects/sec (GRM 3x256+4096 pts, PRM 200x256+200x48 pts)', 'dtype': math, 'data': 'synthetic',
           'objects': objects, 'chunk': chunk}
    gen = torch.Generator().manual_seed(0)
    b = chunk
    nchunks = max(objects // b, 1)

    grm = GeometryTransformer(GCFG, 11, 4).eval()
    grm.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in grm.state_dict().items()}, 1))
    grm = grm.to(dev).set_math(math)
    gd = {'geo_memory_points': torch.randn((b, 4096, 11), generator=gen).to(dev),
          'geo_query_points': torch.randn((b, 3, 256, 4), generator=gen).to(dev),
          'geo_query_boxes': torch.randn((b, 3, 7), generator=gen).to(dev), 'geo_query_num': torch.full((b,), 3)}
    t = timed(lambda: [grm(dict(gd)) for _ in range(nchunks)], steps)
    out['grm_objects_per_s'] = round(nchunks * b / t, 1)
    # algorithmic FLOP per object: encoders 4.4 G + K/V projections 1.1 G
    out['grm_tflops'] = round(nchunks * b * 5.5e9 / t / 1e12, 2)
    del grm, gd

    # the fused kernels that carry the split-math refiner (csrc/pointnet.hip, csrc/mlp_chain.hip), timed inside the PRM pass with events
    # on the launch stream: algorithmic FLOP = 2 * rows * sum(cin * cout) of the layers each fuses
    fused = {}

    def spy(name, fn, flop_of):
        def wrapped(*args, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*args, **kw)
            e1.record()
            fused.setdefault(name, []).append((e0, e1, flop_of(*args, **kw)))
            return r
        return wrapped
    real_pn, real_mc = ops.pointnet3, ops.mlp_chain
    if math != 'f32':
        ops.pointnet3 = spy('k_pointnet3', real_pn, lambda x, layers, *a, **k: 2.0 * x.shape[0] * (32 * 128 + 128 * 128 + 128 * layers[2][0].shape[0]))
        ops.mlp_chain = spy('k_mlp_chain', real_mc, lambda x, la, lb, gs, gr, m, kv=None: 2.0 * x.shape[0] * (128 * 512 + 512 * 256 + (2 * 256 * 256 if kv is not None else 0)))

    prm = PositionTransformer(PCFG, 32, 32).eval()
    prm.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in prm.state_dict().items()}, 2))
    prm = prm.to(dev).set_math(math)
    bp = min(b, 96)
    pchunks = max(objects // bp, 1)
    lens = torch.randint(5, 201, (bp,), generator=gen)
    pad = (torch.arange(200)[None, :] >= lens[:, None]).float()
    pd = {'pos_query_points': torch.randn((bp, 200, 256, 32), generator=gen).to(dev),
          'pos_memory_points': torch.randn((bp, 200, 48, 32), generator=gen).to(dev),
          'pos_trajectory': torch.randn((bp, 200, 7), generator=gen).to(dev), 'padding_mask': pad.to(dev)}
    t = timed(lambda: [prm(dict(pd)) for _ in range(pchunks)], steps)
    out['prm_objects_per_s'] = round(pchunks * bp / t, 1)
    out['prm_tflops'] = round(pchunks * bp * (12.8e9 + 2.5e9 + 2 * 0.98e9) / t / 1e12, 2)
    ops.pointnet3, ops.mlp_chain = real_pn, real_mc
    if fused:
        torch.cuda.synchronize()
        for name, recs in fused.items():
            ms = sum(a.elapsed_time(b) for a, b, _ in recs)
            fl = sum(f for _, _, f in recs)
            out['%s_prm_us' % name] = round(1000.0 * ms / len(recs), 1)
            out['%s_prm_tflops' % name] = round(fl / (ms * 1e-3) / 1e12, 2)
            out['%s_prm_share' % name] = round(ms * 1e-3 / ((1 + steps) * t), 3)         # of the PRM pass's wall time (warm-up call included in both)
    del prm, pd

    # attention core alone, PRM cross-attention shape (200 queries x 9600 keys, 8 heads x 32): exact fp32 on the fp32 matrix cores
    q = torch.randn((bp, 200, 256), generator=gen).to(dev)
    k = torch.randn((bp, 9600, 256), generator=gen).to(dev)
    v = torch.randn((bp, 9600, 256), generator=gen).to(dev)
    t = timed(lambda: ops.mha_core(q, k, v, None, 8, 32 ** -0.5), 5)
    out['mha_core_prm_tflops'] = round(bp * 4.0 * 200 * 9600 * 256 / t / 1e12, 2)
    out['mha_core_prm_us'] = round(t * 1e6, 1)
    del q, k, v

    # GRM cross-attention without projecting the memory (dz_xattn_folded): 3 queries x 4096 memory rows x 256 channels per object; the
    # memory rows are its compulsory HBM bytes (read once)
    if ops.xattn_folded_supported(3, 256, 8):
        q = torch.randn((b, 3, 256), generator=gen).to(dev)
        mem = torch.randn((b, 4096, 256), generator=gen).to(dev)
        wk = (torch.randn((256, 256), generator=gen) / 16).to(dev)
        wv = (torch.randn((256, 256), generator=gen) / 16).to(dev)
        bv = torch.zeros(256, device=dev)
        t = timed(lambda: ops.xattn_folded(q, mem, None, wk, wv, bv, 8, 32 ** -0.5), 10)
        out['xattn_folded_grm_us'] = round(t * 1e6, 1)
        out['xattn_folded_grm_gbs'] = round(b * 4096 * 256 * 4 / t / 1e9, 1)
        del q, mem

    if with_crop:       # object crop mask
        from detzero_amd import roiaware_pool3d_utils
        pts = torch.from_numpy(synth_waymo_frame(0, 180000)[:, :3]).to(dev)[None].contiguous()
        boxes = torch.from_numpy(synth_boxes(0, 128, 60.0)).to(dev)[None].contiguous()
        t = timed(lambda: roiaware_pool3d_utils.points_in_boxes_gpu_v2(pts, boxes), 10)
        out['points_in_boxes_us'] = round(t * 1e6, 1)
        out['points_in_boxes_write_gbs'] = round(128 * 180000 * 4 / t / 1e9, 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--objects', type=int, default=1024)
    ap.add_argument('--chunk', type=int, default=128, help='objects per forward (reference BATCH_SIZE_PER_GPU 128 / 96)')
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--math', default='f32', choices=['f32', 'f16x2', 'bf16x2'], help="arithmetic of the big MLP stacks (set_math)")
    args = ap.parse_args()
    print(json.dumps(measure(torch.device('cuda', 0), args.objects, args.chunk, args.steps, args.math)))


if __name__ == '__main__':
    main()
