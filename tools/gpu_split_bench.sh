#!/bin/bash
# fp32 vs split-precision bench and per-layer sparse timing (development round)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/split
for m in ${@:-f32 f16x2 bf16x2}; do
  echo "==== bench --math $m --batch 4"
  timeout 400 python bench.py --steps 40 --warmup 5 --batch ${BATCH:-4} --math $m --no-cpu-baseline 2> gpurun_out/split/$m.err > gpurun_out/split/$m.json || tail -5 gpurun_out/split/$m.err
  python - $m <<'PY'
import json, sys
m = sys.argv[1]
try:
    d = json.load(open('gpurun_out/split/%s.json' % m))
    print('math', m, 'value', d['value'], 'ms/step', d['ms_per_step'], 'launch', d['config']['launch'][:40], 'conv_ms_per_frame', d.get('conv_ms_per_frame'))
    print('   roofline', d['roofline']['kernel'], d['roofline']['achieved'], '/', d['roofline']['peak'], '=', d['roofline']['frac'])
    for k in d['kernels']:
        print('  %-28s x%-5.1f avg %8.2f us  %7.3f ms/step  %6.2f TF/s' % (k['kernel'], k['launches_per_step'], k['avg_us'], k['ms_per_step'], k['tflops']))
except Exception as e:
    print('no bench json', e)
PY
done
echo "==== per-layer sparse (f16x2)"; timeout 300 python tools/bench_spconv.py --batch ${BATCH:-4} --math f16x2 2>&1 | tail -23
