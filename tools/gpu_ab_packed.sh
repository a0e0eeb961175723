#!/bin/bash
# packed 27-tap tables for the small-channel levels: tests, then A/B inside the detector on ONE box
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -4
B="python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-aux --no-refine --no-pdv --profile-frames 0"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
for r in 1 2; do
  echo -n "packed "; DZ_TUNE_PACKED_TABLES=1 timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"
  echo -n "plain "; timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"
done
echo -n "index packed "; DZ_TUNE_PACKED_TABLES=1 timeout 300 python tools/bench_index.py 2>/dev/null | tail -1 | cut -c35-125
echo -n "index plain "; timeout 300 python tools/bench_index.py 2>/dev/null | tail -1 | cut -c35-125
