#!/bin/bash
# batch x chains sweep of bench.py (f16x2); third field "g" = graph replay, "e" = eager
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/sweep
for bc in "${@}"; do
  set -- $bc
  extra=""; [ "$3" = "e" ] && extra="--no-graph"
  timeout 400 python bench.py --steps 40 --warmup 5 --batch $1 --chains $2 $extra --no-cpu-baseline --profile-frames 0 2> gpurun_out/sweep/c.err > gpurun_out/sweep/c.json || tail -3 gpurun_out/sweep/c.err
  python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/sweep/c.json'))
    print('batch', d['config']['frames_per_step_per_gpu'], 'chains', d['config']['chains'], 'value', d['value'], 'ms/step', d['ms_per_step'], d['config']['launch'][:60])
except Exception as e:
    print('no bench json', e)
PY
done
