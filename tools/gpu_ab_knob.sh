#!/bin/bash
# GPU test suite, then the headline with / without environment knobs, two alternating rounds on ONE box:
#   tools/gpu_ab_knob.sh "NAME=VALUE [NAME2=VALUE2 ...]" [notests]
# e.g. DZ_TUNE_EAGER_PYRAMID=1, DZ_BEV_SCATTER=1, DZ_TUNE_PACKED_TABLES=1, "DZ_TUNE_NBR_GENERIC=1 DZ_TUNE_MARK_PLAIN=1 DZ_TUNE_LINE_FLAGS=0"
cd $GRAFT_REPO_ROOT
if [ -z "$2" ]; then timeout 1200 python -m pytest tests -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -4; fi
B="python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-aux --no-refine --no-pdv --profile-frames 0"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
for r in 1 2; do
  echo -n "default "; timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"
  echo -n "$1 "; env $1 timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"
done
