#!/bin/bash
# x-run engine: its parity tests, the per-layer timing of both engines, and the headline A/B on ONE box.
#   tools/gpu_xrun.sh [notests] [nobench]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ "$1" != "notests" ]; then
  timeout 900 python -m pytest tests/test_gpu_xrun.py -q -x --timeout=600 -p no:cacheprovider -s 2>&1 | tail -25 | tee gpurun_out/xrun_tests.txt
fi
for e in gather xrun; do
  echo "== bench_spconv engine $e"
  DZ_TUNE_SPCONV_ENGINE=$e timeout 300 python tools/bench_spconv.py --batch 16 --reps 20 --math f16x2 2>&1 | tail -24 | tee gpurun_out/xrun_spconv_$e.txt
done
if [ "$2" != "nobench" ]; then
  B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-aux --no-refine --no-pdv --profile-frames 0"
  P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
  for r in 1 2; do
    for e in gather xrun; do
      echo -n "$e "; timeout 300 $B --sparse-engine $e 2>/dev/null | tail -1 | python -c "$P"
    done
  done | tee gpurun_out/xrun_ab.txt
fi
