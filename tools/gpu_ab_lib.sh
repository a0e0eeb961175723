#!/bin/bash
# A/B of two builds of the library on ONE box: abtest/lib_old.so vs abtest/lib_new.so, alternating
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in old new; do
  cp abtest/lib_$v.so detzero_amd/libdetzero_hip.so
  echo -n "$v "; timeout 300 python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-aux --profile-frames 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
