#!/bin/bash
# A/B of one environment knob inside the detector on ONE box: tools/gpu_ab_env.sh NAME VALUE_A VALUE_B  (alternating, three rounds)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in $2 $3; do
  echo -n "$1=$v "; env $1=$v timeout 300 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-aux --profile-frames 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
