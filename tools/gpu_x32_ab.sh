#!/bin/bash
# 32-channel x-run configurations (DZ_TUNE_X32 = 0 / 1 / 2): parity tests, per-layer timing, headline.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 0 1 2; do
  echo "== DZ_TUNE_X32=$v"
  DZ_TUNE_X32=$v timeout 300 python -m pytest tests/test_gpu_xrun.py -q -x --timeout=120 -p no:cacheprovider -k "32 or detector" 2>&1 | tail -2
  DZ_TUNE_X32=$v timeout 200 python tools/bench_spconv.py --batch 16 --reps 20 --math f16x2 --only 32-32 2>&1 | grep -E "^x" | sort -u | cut -c1-30,95-150
done 2>&1 | tee gpurun_out/x32_ab.txt
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-aux --no-refine --no-pdv --profile-frames 0"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
for r in 1 2; do for v in 0 1 2; do echo -n "x32=$v "; DZ_TUNE_X32=$v timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"; done; done | tee -a gpurun_out/x32_ab.txt
