#!/bin/bash
# sparse conv development round: parity tests, per-layer times, bench line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
echo "==== tests"; timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -x --timeout=600 2>&1 | tail -8
timeout 200 python tools/bench_spconv.py --batch ${BATCH:-16} --math f16x2 2>&1 | grep "^k\|^sum" | grep -v "+res" | uniq 
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
