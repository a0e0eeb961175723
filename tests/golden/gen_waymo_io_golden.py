"""Golden vectors for the frame assembly (sweep selection + merge) from the REFERENCE's own code, CPU.

    python tests/golden/gen_waymo_io_golden.py      (build container only; needs /root/reference)

DatasetTemplate.get_sweep_idxs and DatasetTemplate.merge_sweeps (detection/detzero_det/datasets/dataset.py:140-195) are
static methods; importing the module pulls in the data processors and spconv, so their source is extracted with `ast`
and executed as is (globals: numpy only) on seeded synthetic sweeps with Waymo-like world poses.
"""
import ast
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)

CASES = [(1, [0, 0], 3), (2, [-2, 0], 5), (3, [-1, 1], 0), (4, [-2, 2], 9)]      # (seed, SWEEP_COUNT, index of the current frame) in a 10-frame sequence


def reference_functions():
    src = open(REF + '/detection/detzero_det/datasets/dataset.py').read()
    fns = {}
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.FunctionDef) and node.name in ('get_sweep_idxs', 'merge_sweeps'):
            node.decorator_list = []
            glb = {'np': np}
            exec(compile(ast.Module(body=[node], type_ignores=[]), 'dataset.py:' + node.name, 'exec'), glb)
            fns[node.name] = glb[node.name]
    return fns['get_sweep_idxs'], fns['merge_sweeps']


def synth_sequence(seed, n_frames=10, n_points=3000):
    """Infos and raw sweeps of one sequence: vehicle driving along a curve, 10 Hz, 6-column frames with ~7 % NLZ returns."""
    from detzero_amd.synth import synth_waymo_frame
    rng = np.random.default_rng(seed)
    infos, sweeps = [], []
    pos, yaw = np.array([8000.0, -2500.0, 30.0]) + rng.uniform(-100, 100, 3), rng.uniform(-np.pi, np.pi)
    t0 = 1550000000000000 + int(rng.integers(0, 10 ** 9))
    for i in range(n_frames):
        yaw += rng.normal(0.01, 0.005)
        pos = pos + 0.9 * np.array([np.cos(yaw), np.sin(yaw), 0.002])
        pose = np.eye(4)
        pose[:2, :2] = [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]
        pose[:3, 3] = pos
        p5 = synth_waymo_frame(seed * 100 + i, n_points=n_points)
        p5[:, 3] = rng.uniform(0, 4, size=n_points)                      # raw intensity (before tanh)
        nlz = np.where(rng.random(n_points) < 0.07, 1.0, -1.0).astype(np.float32)
        sweeps.append(np.concatenate([p5, nlz[:, None]], axis=1).astype(np.float32))
        infos.append({'sample_idx': i, 'sequence_len': n_frames, 'sequence_name': 'seq%d' % seed, 'pose': pose,
                      'time_stamp': t0 + i * 100000 + int(rng.integers(-300, 300))})
    return infos, sweeps


def main():
    get_sweep_idxs, merge_sweeps = reference_functions()
    out = {}
    for seed, sweep_count, idx in CASES:
        infos, sweeps = synth_sequence(seed)
        tl = get_sweep_idxs(infos[idx], sweep_count, idx)
        merged = merge_sweeps(infos[idx], [infos[i] for i in tl], [sweeps[i].copy() for i in tl])
        out['c%d_idx' % seed] = np.asarray(tl)
        out['c%d_points' % seed] = merged
        print(seed, sweep_count, idx, '->', list(tl), merged.shape, merged.dtype)
    np.savez_compressed(os.path.join(HERE, 'waymo_io_golden.npz'), **out)
    print('saved', os.path.getsize(os.path.join(HERE, 'waymo_io_golden.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
