#!/bin/bash
# GPU test suite, then the headline with / without environment knobs, two alternating rounds on ONE box:
#   tools/gpu_ab_knob.sh "NAME=VALUE [NAME2=VALUE2 ...]" [notests] [layers]
# e.g. DZ_TUNE_EAGER_PYRAMID=1, DZ_BEV_SCATTER=1, DZ_TUNE_XRUN_SORT=0, DZ_TUNE_X32=1, DZ_TUNE_SPCONV_ENGINE=gather, "--batch 32" style
# bench arguments go through BENCH_ARGS.  layers: also the per-layer sparse-convolution table (tools/bench_spconv.py) under both settings.
cd $GRAFT_REPO_ROOT
if [ -z "$2" ] || [ "$2" = "-" ]; then timeout 1200 python -m pytest tests -m gpu -q -x --timeout=600 -p no:cacheprovider 2>&1 | tail -4; fi
if [ "$3" = "layers" ]; then
  for k in "" "$1"; do
    echo "== per-layer sparse convolutions [${k:-default}]"
    env $k timeout 300 python tools/bench_spconv.py --batch 16 --reps 20 --math f16x2 2>&1 | grep -E "^x|^g|^sum" | sort -u | cut -c1-30,95-150
  done
fi
B="python bench.py --steps 150 --warmup 10 --no-cpu-baseline --no-aux --no-refine --no-pdv --profile-frames 0 $BENCH_ARGS"
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
for r in 1 2; do
  echo -n "default "; timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"
  echo -n "$1 "; env $1 timeout 300 $B 2>/dev/null | tail -1 | python -c "$P"
done
