#!/bin/bash
# tap-set order A/B on ONE box: x-run parity tests, per-layer timing and the headline with DZ_TUNE_XRUN_SORT=0/1.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_xrun.py -q -x --timeout=300 -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/sort_tests.txt
for s in 0 1; do
  echo "== bench_spconv sort $s"
  DZ_TUNE_XRUN_SORT=$s timeout 300 python tools/bench_spconv.py --batch 16 --reps 20 --math f16x2 --only 32-32,64-64,128-128 2>&1 | tail -14 | tee gpurun_out/sort_spconv_$s.txt
done
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-aux --no-refine --no-pdv --profile-frames 0"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["stages"]["sparse_backbone"] if "stages" in d else "")'
for r in 1 2; do
  for s in 0 1; do
    echo -n "sort$s "; DZ_TUNE_XRUN_SORT=$s timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"
  done
done | tee gpurun_out/sort_ab.txt
