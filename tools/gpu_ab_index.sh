#!/bin/bash
# the index-chain kernels of round 3 inside the detector, A/B on ONE box (old kernels through their development knobs)
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-aux --no-refine --no-pdv --profile-frames 0"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
for r in 1 2; do
  echo -n "new "; timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"
  echo -n "old-knobs "; DZ_TUNE_NBR_GENERIC=1 DZ_TUNE_MARK_PLAIN=1 DZ_TUNE_LINE_FLAGS=0 timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"
done
