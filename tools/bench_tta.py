"""Timing of the TTA merge on one GPU (development / DESIGN.md numbers): restore + weighted box fusion of 15 copies x up to
500 boxes for a batch of frames, synthetic detections with realistic agreement between the copies."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import gen_tta_golden as gen          # noqa: E402
from detzero_amd import tta           # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    aug = tta.TestTimeAugmentor(gen.AUG_CONFIG)
    ops, t = aug.op_names, len(aug.op_names)
    frames, m = 16, 500
    pb, ps, pl = np.zeros((frames, t, m, 7), np.float32), np.zeros((frames, t, m), np.float32), np.zeros((frames, t, m), np.int32)
    for f in range(frames):
        preds = gen.synth_predictions(100 + f, 420, ops)
        for i, p in enumerate(preds):
            n = min(len(p['pred_boxes']), m)
            pb[f, i, :n], ps[f, i, :n], pl[f, i, :n] = p['pred_boxes'].numpy()[:n], p['pred_scores'].numpy()[:n], p['pred_labels'].numpy()[:n]
    boxes, scores, labels = torch.from_numpy(pb).to(dev), torch.from_numpy(ps).to(dev).reshape(frames, t * m), torch.from_numpy(pl).to(dev).reshape(frames, t * m)

    def run():
        b = tta.restore_boxes(boxes.clone(), ops)
        return tta.wbf_fuse_nosync(b.reshape(frames, t * m, 7), scores, labels, t)
    out = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    cand = int((labels > 0).sum().item())
    print('TTA merge: %d frames x %d copies, %d candidate boxes -> %s fused; restore + fusion %.2f ms per batch (%.2f ms per frame)'
          % (frames, t, cand, out[3].tolist()[:4], ms, ms / frames))
    pts = torch.from_numpy(np.random.default_rng(0).uniform(-70, 70, size=(160000, 5)).astype(np.float32)).to(dev)
    aug.augment(pts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        aug.augment(pts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print('dz_tta_augment_points: 160k points -> %d copies in %.3f ms (%.2f TB/s of copy traffic)' % (t, dt * 1e3, (t + 1) * 160000 * 20 / dt / 1e12))


if __name__ == '__main__':
    main()
