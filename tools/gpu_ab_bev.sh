#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_full_parity.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
B="python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-aux --no-refine --no-pdv --profile-frames 0"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
for r in 1 2; do
  echo -n "dense "; timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"
  echo -n "scatter "; DZ_BEV_SCATTER=1 timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"
done
