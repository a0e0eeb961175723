#!/bin/bash
# bench line + eager kernel trace summary (development)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-220
rm -rf gpurun_out/prof/trace
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline --profile-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/trace_stdout.txt 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof/trace/bench_results.db > gpurun_out/prof/trace_summary.txt; head -${LINES_OUT:-26} gpurun_out/prof/trace_summary.txt | cut -c1-150
find gpurun_out/prof -name "*.db" -delete
