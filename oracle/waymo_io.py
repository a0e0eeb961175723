"""ORACLE (test infrastructure only): CPU restatement of the frame assembly of the detector's dataset,
detection/detzero_det/datasets/dataset.py:140-195 (DatasetTemplate.get_sweep_idxs, merge_sweeps).  Pinned by
tests/golden/waymo_io_golden.npz (gen_waymo_io_golden.py runs the reference's two static methods, extracted from the
file with `ast`, on seeded synthetic sweeps)."""
import numpy as np


def get_sweep_idxs(current_info, sweep_count=(0, 0), current_idx=0):
    """dataset.py:141-162: indices (into the info list) of the frames current+lo .. current+hi, clamped to the sequence."""
    cur, n = current_info['sample_idx'], current_info['sequence_len']
    want = cur + np.arange(sweep_count[0], sweep_count[1] + 1)
    want = np.clip(want, 0, n - 1)
    return current_idx + (want - cur)


def merge_sweeps(info, target_infos, points):
    """dataset.py:164-195: (N_i,6) float32 [x,y,z,intensity,elongation,NLZ] per sweep -> (N',6) float64
    [x,y,z (current frame, float32 values), tanh(intensity), elongation, time offset in s]."""
    out = []
    for tinfo, p in zip(target_infos, points):
        q = p[:, 0:5][p[:, 5] == -1]
        q[:, 3] = np.tanh(q[:, 3])
        mat = np.linalg.inv(info['pose']) @ tinfo['pose']
        dt = int(tinfo['time_stamp']) - int(info['time_stamp'])
        q[:, :3] = np.concatenate([q[:, :3], np.ones((q.shape[0], 1))], axis=1) @ mat[:3, :].T
        out.append(np.concatenate([q, float(dt) / 1000000. * np.ones((q.shape[0], 1))], axis=1))
    return np.concatenate(out, axis=0)
