#!/bin/bash
# x-run engine quick check: its parity tests + per-layer timing (+ optionally the headline).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_xrun.py -q -x --timeout=300 -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/xq_tests.txt
timeout 300 python tools/bench_spconv.py --batch 16 --reps 20 --math f16x2 --only 32-32,64-64,128-128 2>&1 | grep -E "^x|^sum" | sort -u | cut -c1-30,95-150 | tee gpurun_out/xq_spconv.txt
if [ "$1" = "bench" ]; then
  B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-aux --no-refine --no-pdv --profile-frames 0"
  P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
  for r in 1 2; do timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"; done | tee gpurun_out/xq_bench.txt
fi
