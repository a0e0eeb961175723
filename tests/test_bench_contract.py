"""The bench line's contract, checked on the newest committed line (profiles/*_bench_graph.json) and on bench.py's argument surface
(no GPU needed): every key the driver and the judge read is there, with the types and relations the task states."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_line():
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_graph.json')))
    assert files, 'no committed bench line under profiles/'
    with open(files[-1]) as f:
        return json.loads(f.read().strip().splitlines()[-1]), files[-1]


def test_committed_bench_line_has_the_contract_keys():
    d, path = _newest_line()
    for k, t in (('metric', str), ('value', float), ('unit', str), ('n_gpus', int), ('steps', int), ('warmup', int),
                 ('ms_per_step', float), ('higher_is_better', bool), ('scaling', str), ('dtype', str), ('data', str), ('config', dict)):
        assert isinstance(d[k], t), (path, k)
    assert 'vs_baseline' in d and d['vs_baseline'] is None            # BASELINE.md holds no published number for this metric
    assert d['scaling'] == 'weak' and d['data'] == 'synthetic' and d['higher_is_better'] is True
    assert 'workload' in d['config'] and 'model' not in d['config']
    per_step = d['config']['frames_per_step_per_gpu']
    assert abs(d['value'] - d['n_gpus'] * per_step * 1000.0 / d['ms_per_step']) / d['value'] < 1e-3      # value = frames / time
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and 0 < r['frac'] <= 1
    # HBM bytes of the dominant kernel per launch from the PMC passes (r04: a number, its read / write split beside it; r01-r03 lines: a dict)
    t = r['traffic']
    assert t is None or (t['bytes'] > 0 if isinstance(t, dict) else (t > 0 and abs(t - r['traffic_read_bytes'] - r['traffic_write_bytes']) <= 2))
    if 'hbm' in r:                                                        # the north star's stage fractions (r04)
        for st in ('voxelize', 'index', 'sparse_backbone', 'voxelize_plus_backbone'):
            h = r['hbm'][st]
            assert h['bytes'] > 0 and h['ms'] > 0 and abs(h['gbs'] - h['bytes'] / h['ms'] / 1e6) / h['gbs'] < 1e-2
            assert abs(h['frac'] - h['gbs'] / r['hbm_peak_gbs']) < 1e-3
    if 'hbm_sparse_backbone_frac' in r and 'hbm' not in r:              # r05: flat keys only, the four fractions right behind the contract keys
        keys = list(r)
        assert keys[:6] == ['bound', 'achieved', 'peak', 'unit', 'frac', 'traffic']
        assert keys[6:10] == ['hbm_voxelize_plus_backbone_frac', 'hbm_sparse_backbone_frac', 'hbm_index_frac', 'hbm_voxelize_frac']
        for st in ('voxelize', 'index', 'sparse_backbone', 'voxelize_plus_backbone'):
            b, ms, gbs, fr = (r['hbm_%s_%s' % (st, k)] for k in ('bytes', 'ms', 'gbs', 'frac'))
            assert b > 0 and ms > 0 and abs(gbs - b / ms / 1e6) / gbs < 1e-2 and abs(fr - gbs / r['hbm_peak_gbs']) < 1e-3
        assert d['config']['latency_ms_per_pass'] == d['ms_per_step'] and 'ref_batch' in d['config']['like_for_like']
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['value'] > 0 and c['cores'] >= 1 and isinstance(c['sample'], str)
    assert c['unit'] == d['unit']


def test_committed_bench_line_has_the_auxiliary_legs():
    """Round 2: what the headline does not show rides in the same line - an exact-fp32 run with its own roofline, the
    reference's batch size, ragged frames, host-resident frames, and per-stage HBM figures (north star: HBM GB/s on the scatter)."""
    d, path = _newest_line()
    if 'fp32' not in d:
        import pytest
        pytest.skip('line predates the auxiliary legs')
    f = d['fp32']
    assert f['math'] == 'f32' and f['dtype'] == 'f32' and f['value'] > 0 and f['unit'] == d['unit']
    assert abs(f['value'] - f['frames_per_step'] * 1000.0 / f['ms_per_step']) / f['value'] < 1e-3
    r = f['roofline']
    assert r['bound'] == 'mfma' and abs(r['peak'] - 157.3) < 0.1 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert d['ref_batch']['frames_per_step'] == 8 and d['ref_batch']['value'] > 0
    for k in ('padded', 'list'):
        leg = d['ragged'][k]
        assert leg['value'] > 0 and 150000 <= leg['mean_points_per_frame'] <= 180000
    if 'f16' in d:                                                          # opt-in single-product mode: never the headline
        assert d['f16']['math'] == 'f16' and d['f16']['value'] > 0 and d['dtype'] != d['f16']['dtype']
    if 'multisweep' in d:                                                   # BASELINE configs[4] shape (later lines of round 2)
        ms = d['multisweep']
        assert ms['value'] > 0 and ms['mean_points_per_frame'] == 320000 and ms['frames_per_step'] == 8
    h = d['with_h2d']
    assert h['value'] > 0 and h['h2d_bytes_per_step'] == h['frames_per_step'] * 160000 * 5 * 4
    assert h['value'] <= d['value'] * 1.05                                  # the H2D-inclusive rate is never the headline
    stages = {s['stage']: s for s in d['stages']}
    assert list(stages) == ['voxelize', 'index', 'sparse_backbone', 'dense', 'post']
    for k in ('voxelize', 'index', 'sparse_backbone'):
        s = stages[k]
        assert s['algorithmic_bytes'] > 0 and abs(s['hbm_gbs'] - s['algorithmic_bytes'] / (s['ms_per_step'] * 1e-3) / 1e9) / s['hbm_gbs'] < 1e-2
        assert abs(s['frac_of_hbm_peak'] - s['hbm_gbs'] / 8000.0) < 1e-3


def test_bench_arguments():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--help'], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ('--gpus', '--steps', '--warmup'):
        assert flag in out.stdout


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` outside a launcher re-executes under torch.distributed.run and really runs two ranks: the --stub
    switch replaces the GPU work by a no-op step (gloo on CPU) but keeps the spawn, the barrier + gather + max-over-ranks timed
    region and the JSON line - n_gpus and the all-reduced ranks_seen are 2, and rank 0 holds both ranks' gathered counts."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--stub'],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout                                   # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['ranks_seen'] == 2 and d['steps'] == 3 and d['warmup'] == 1
    assert d['config']['gathered_counts'] == [1, 2]                      # rank r wrote r + 1: the gather reached rank 0
    assert d['data'].startswith('stub')


def test_bench_refuses_more_gpus_than_the_node_has():
    """No silent single-rank fallback: without GPUs (this container) --gpus 2 must fail, not print a 1-GPU line."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip('node has the GPUs')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and 'GPU' in (out.stderr + out.stdout)
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
