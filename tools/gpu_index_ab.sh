#!/bin/bash
# A/B of the index-chain kernels on one box: bench_index with each development knob set back to the old kernel.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $O
if [ -n "$1" ]; then timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -x 2>&1 | tail -5; fi
run() { echo "== $*"; env "$@" timeout 300 python tools/bench_index.py 2>/dev/null | tail -1 | cut -c1-140; }
run A=0
run DZ_TUNE_NBR_FLAT=1
run DZ_TUNE_NBR_GENERIC=1 DZ_TUNE_MARK_PLAIN=1 DZ_TUNE_LINE_FLAGS=0
run A=0
bash tools/gpu_index.sh ${2:-idx3} | tail -4
