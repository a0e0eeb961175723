#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
rm -rf gpurun_out/prof/traceg
cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof/traceg -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-frames 0 ${BENCH_ARGS} > $GRAFT_REPO_ROOT/gpurun_out/prof/traceg_stdout.txt 2>&1; cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/prof/traceg_stdout.txt | cut -c1-200
python tools/trace_timeline.py gpurun_out/prof/traceg/bench_results.db 2 > gpurun_out/prof/traceg_timeline.txt
tail -1 gpurun_out/prof/traceg_timeline.txt
find gpurun_out/prof -name "*.db" -delete
