#!/bin/bash
# Frames-per-step sweep of bench.py on one GPU (graph replay), plus the batched-vs-single parity test.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/sweep
echo "==== batched parity test"; timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "batched or pipeline" --timeout=600 2>&1 | tail -5
for b in ${@:-1 2 4 8}; do
  echo "==== bench --batch $b"
  timeout 400 python bench.py --steps 40 --warmup 5 --batch $b --no-cpu-baseline 2> gpurun_out/sweep/b$b.err > gpurun_out/sweep/b$b.json || tail -5 gpurun_out/sweep/b$b.err
  python - $b <<'PY'
import json, sys
b = sys.argv[1]
try:
    d = json.load(open('gpurun_out/sweep/b%s.json' % b))
    print('batch', b, 'value', d['value'], 'ms/step', d['ms_per_step'], 'launch', d['config']['launch'][:40], 'conv_ms_per_frame', d.get('conv_ms_per_frame'))
    for k in d['kernels']:
        print('  %-28s x%-5.1f avg %8.2f us  %7.3f ms/step  %6.2f TF/s' % (k['kernel'], k['launches_per_step'], k['avg_us'], k['ms_per_step'], k['tflops']))
except Exception as e:
    print('no bench json', e)
PY
done
