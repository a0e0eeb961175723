#!/bin/bash
# three-way alternating A/B of environment settings on ONE box: tools/gpu_ab3.sh "<env A>" "<env B>" "<env C>" [rounds]   (BENCH_ARGS as in gpu_ab_knob.sh)
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 120 --warmup 10 --no-cpu-baseline --no-aux --no-refine --no-pdv --profile-frames 0 $BENCH_ARGS"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
for r in $(seq 1 ${4:-3}); do
  for k in "$1" "$2" "$3"; do
    echo -n "[${k:-default}] "; env $k timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"
  done
done
