#!/bin/bash
# Frames-per-step sweep of bench.py on one GPU (graph replay, default math)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/sweep
for b in ${@:-1 2 4 8 16}; do
  timeout 400 python bench.py --steps 30 --warmup 5 --batch $b --no-cpu-baseline --profile-frames 0 2> gpurun_out/sweep/b$b.err > gpurun_out/sweep/b$b.json || tail -5 gpurun_out/sweep/b$b.err
  python - $b <<'PY'
import json, sys
b = sys.argv[1]
try:
    d = json.load(open('gpurun_out/sweep/b%s.json' % b))
    print('batch', b, 'value', d['value'], 'ms/step', d['ms_per_step'], 'ms/frame', d['config']['ms_per_frame'], d['config']['launch'][:40])
except Exception as e:
    print('no bench json', e)
PY
done
