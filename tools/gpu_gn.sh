#!/bin/bash
# register-ring sparse conv (tile masks from build_neighbors) vs the LDS-table variant: parity tests, per-layer times, bench
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/gn
echo "==== tests"; timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -x --timeout=600 2>&1 | tail -8
for g in 0 1; do
  echo "== NOGN=$g"
  DZ_TUNE_SPCONV_NOGN=$g timeout 200 python tools/bench_spconv.py --batch ${BATCH:-16} --math f16x2 2>&1 | grep "^k\|^sum" | grep -v "+res" | uniq
  DZ_TUNE_SPCONV_NOGN=$g timeout 300 python bench.py --steps 30 --warmup 5 2>&1 | tail -1 | cut -c1-200
done
