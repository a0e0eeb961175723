#!/bin/bash
# bench_index under values of one development knob: tools/gpu_knob_index.sh NAME v1 v2 ...
cd "$(dirname "$0")/.."
N=$1; shift
for v in "$@"; do echo -n "$N=$v "; env $N=$v timeout 300 python tools/bench_index.py 2>/dev/null | tail -1 | cut -c35-125; done
